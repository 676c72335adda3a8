// ref_sources_check.cpp -- the REFERENCE's own EdgeSE2RhoTheta::computeError (src/sparse_gslam/src/g2o_bindings/
// edge_se2_rhotheta.cpp:9-16, through ls_extractor/utils.h:22-45), compiled verbatim from the read-only checkout and linked
// in, against this repo's restatement of it (tests/cpp/landmark_rhotheta.cpp, oracle/np_lm_oracle.py) on 2000 pose / line
// pairs that cover the wraps and the sign flip of rho: the two must agree to rounding.  Built and run by
// tests/test_landmark_lm.py where the checkout is present.
#include <iostream>
#include <iomanip>
#include "g2o/core/base_binary_edge.h"
#include "g2o/core/base_vertex.h"
#include "g2o/types/slam2d/vertex_se2.h"
#include "g2o/stuff/misc.h"
#include "g2o_bindings/edge_se2_rhotheta.h"
#include "g2o_bindings/vertex_rhotheta.h"
using namespace g2o;
inline Eigen::Vector2d move_line(const Eigen::Vector2d& rt, const Eigen::Vector2d& trans, double angle) {
  double th = rt[1] + angle;
  if (th > const_pi()) th -= 2 * const_pi();
  if (th < -const_pi()) th += 2 * const_pi();
  double rho = rt[0] + trans[0] * std::cos(th) + trans[1] * std::sin(th);
  if (rho < 0.0) { rho = -rho; th += const_pi(); if (th > const_pi()) th -= 2 * const_pi(); }
  return Eigen::Vector2d(rho, th);
}
int main() {
  std::cout << std::setprecision(17);
  double worst = 0;
  for (int k = 0; k < 2000; ++k) {
    VertexSE2 p; VertexRhoTheta l; EdgeSE2RhoTheta e;
    double x = std::sin(k * 1.3) * 5, y = std::cos(k * 0.7) * 5, t = std::sin(k * 2.1) * 3.1;
    double r = 1 + std::fabs(std::sin(k * 0.37)) * 6, th = std::sin(k * 0.91) * 3.1;
    p.setEstimate(SE2(x, y, t)); l.setEstimate(Eigen::Vector2d(r, th));
    e.vertices()[0] = &p; e.vertices()[1] = &l; e.setMeasurement(Eigen::Vector2d(2.0, 0.3));
    e.computeError();
    const SE2 pinv = p.estimate().inverse();
    Eigen::Vector2d pred = move_line(l.estimate(), pinv.translation(), pinv.rotation().angle());
    double e0 = 2.0 - pred[0], e1 = normalize_theta(0.3 - pred[1]);
    double d = std::max(std::fabs(e0 - e.error()[0]), std::fabs(e1 - e.error()[1]));
    if (d > worst) worst = d;
  }
  std::cout << worst << "\n";
  return worst <= 1e-13 ? 0 : 1;
}
