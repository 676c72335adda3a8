// landmark_graph.cpp -- the non-hot-path half of the g2o surface sparse-gslam uses: a graph with a
// user-defined 2-dof vertex and a user-defined binary edge that only implements computeError()
// (numeric Jacobians), optimised with Levenberg-Marquardt and driven with push()/pop()/discardTop()
// and updateInitialization() the way src/sparse_gslam/src/drone.cpp:146-187 drives its landmark
// graph (solver stack as src/sparse_gslam/src/graphs.cpp:9-15).  The vertex / edge classes below
// are this test's own (a line in Hesse normal form seen from a pose); they only mirror the SHAPE of
// include/g2o_bindings/{vertex_rhotheta,edge_se2_rhotheta}.h: BaseVertex<2, Vector2d> with an
// additive oplus and BaseBinaryEdge<2, Vector2d, VertexSE2, ...> without linearizeOplus.
//
// Runs on the host solver of include/g2o/sgo_g2o_compat.h (no GPU needed).  Prints
//   chi2_before chi2_after iterations max_pose_err max_line_err chi2_after_bad_pop
#include <cmath>
#include <deque>
#include <iomanip>
#include <iostream>

#include "g2o/core/base_binary_edge.h"
#include "g2o/core/base_vertex.h"
#include "g2o/core/block_solver.h"
#include "g2o/core/factory.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/solvers/eigen/linear_solver_eigen.h"
#include "g2o/stuff/macros.h"
#include "g2o/stuff/misc.h"
#include "g2o/types/slam2d/edge_se2.h"
#include "g2o/types/slam2d/vertex_se2.h"

namespace g2o {

class VertexLine2 : public BaseVertex<2, Eigen::Vector2d> {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW;
  VertexLine2() = default;
  void setToOriginImpl() override { _estimate.setZero(); }
  void oplusImpl(const double* u) override {
    _estimate[0] += u[0];
    _estimate[1] += u[1];
  }
  bool read(std::istream&) override { return true; }
  bool write(std::ostream& os) const override { return os.good(); }
};

// (rho, theta) of world line l as seen from pose X = (t, phi)
inline Eigen::Vector2d line_in_frame(const Eigen::Vector2d& l, const SE2& X) {
  double th = normalize_theta(l[1] - X.rotation().angle());
  double rho = l[0] - (X.translation()[0] * std::cos(l[1]) + X.translation()[1] * std::sin(l[1]));
  if (rho < 0) {
    rho = -rho;
    th = normalize_theta(th + const_pi());
  }
  return Eigen::Vector2d(rho, th);
}

class EdgePoseLine : public BaseBinaryEdge<2, Eigen::Vector2d, VertexSE2, VertexLine2> {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW;
  EdgePoseLine() = default;
  void computeError() override {
    const auto* pose = static_cast<VertexSE2*>(_vertices[0]);
    const auto* line = static_cast<VertexLine2*>(_vertices[1]);
    Eigen::Vector2d pred = line_in_frame(line->estimate(), pose->estimate());
    _error[0] = _measurement[0] - pred[0];
    _error[1] = normalize_theta(_measurement[1] - pred[1]);
  }
  bool read(std::istream&) override { return true; }
  bool write(std::ostream& os) const override { return os.good(); }
};
G2O_REGISTER_TYPE(VERTEX_LINE2, VertexLine2);
G2O_REGISTER_TYPE(EDGE_POSE_LINE, EdgePoseLine);

}  // namespace g2o

int main() {
  using namespace g2o;
  using SlamBlockSolver = BlockSolver<BlockSolverTraits<-1, 2>>;
  using SlamLinearSolver = LinearSolverEigen<SlamBlockSolver::PoseMatrixType>;
  SparseOptimizer opt;
  opt.setAlgorithm(new OptimizationAlgorithmLevenberg(g2o::make_unique<SlamBlockSolver>(g2o::make_unique<SlamLinearSolver>())));
  opt.setVerbose(false);
  opt.setComputeBatchStatistics(false);

  const int NP = 8, NL = 5;
  std::deque<VertexSE2> poses(NP);
  std::deque<EdgeSE2> odom(NP - 1);
  std::deque<VertexLine2> lines(NL);
  std::deque<EdgePoseLine> obs;
  SE2 truth[NP];
  Eigen::Vector2d ltruth[NL] = {{4.0, 0.3}, {6.0, 1.7}, {3.0, -2.0}, {8.0, 2.9}, {5.0, -0.9}};
  for (int k = 0; k < NP; ++k) truth[k] = SE2(0.6 * k, 0.2 * std::sin(0.7 * k), 0.15 * k);

  HyperGraph::VertexSet vset;
  HyperGraph::EdgeSet eset;
  for (int k = 0; k < NP; ++k) {
    poses[k].setId(k);
    SE2 init = truth[k];
    if (k > 0) init = SE2(truth[k][0] + 0.05 * std::cos(3.0 * k), truth[k][1] - 0.04 * std::sin(2.0 * k), truth[k][2] + 0.03 * std::cos(5.0 * k));
    poses[k].setEstimate(init);
    if (k == 0) poses[k].setFixed(true);
    opt.addVertex(&poses[k]);
    vset.insert(&poses[k]);
  }
  for (int k = 0; k + 1 < NP; ++k) {
    odom[k].vertices()[0] = &poses[k];
    odom[k].vertices()[1] = &poses[k + 1];
    odom[k].setMeasurement(truth[k].inverse() * truth[k + 1]);
    Eigen::Matrix3d cov;
    cov << 0.01, 0.001, 0, 0.001, 0.02, 0, 0, 0, 0.005;
    odom[k].information().noalias() = cov.inverse();
    opt.addEdge(&odom[k]);
    eset.insert(&odom[k]);
  }
  for (int j = 0; j < NL; ++j) {
    lines[j].setId(10000000 + j);   // landmark ids start at 1e7 in the reference (drone.h:22)
    lines[j].setEstimate(Eigen::Vector2d(ltruth[j][0] + 0.07 * std::cos(1.0 + j), ltruth[j][1] - 0.04 * std::sin(2.0 + j)));
    opt.addVertex(&lines[j]);
    vset.insert(&lines[j]);
  }
  for (int k = 0; k < NP; ++k)
    for (int j = 0; j < NL; ++j) {
      if ((k + j) % 2) continue;
      obs.emplace_back();
      auto& e = obs.back();
      e.vertices()[0] = &poses[k];
      e.vertices()[1] = &lines[j];
      e.setMeasurement(line_in_frame(ltruth[j], truth[k]));
      Eigen::Matrix2d cov;
      cov << 0.004, 0.0005, 0.0005, 0.002;
      e.information().noalias() = cov.inverse();
      opt.addEdge(&e);
      eset.insert(&e);
    }

  opt.initializeOptimization();
  opt.computeActiveErrors();
  const double chi2_before = opt.activeChi2();
  int dof = 0;
  for (auto* edge : opt.activeEdges()) dof += static_cast<OptimizableGraph::Edge*>(edge)->dimension();
  opt.push();
  const int its = opt.optimize(15, false);
  opt.computeActiveErrors();
  const double chi2_after = opt.activeChi2();
  opt.discardTop();

  double perr = 0, lerr = 0;
  for (int k = 0; k < NP; ++k)
    for (int q = 0; q < 3; ++q) perr = std::max(perr, std::fabs(poses[k].estimate()[q] - truth[k][q]));
  for (int j = 0; j < NL; ++j)
    for (int q = 0; q < 2; ++q) lerr = std::max(lerr, std::fabs(lines[j].estimate()[q] - ltruth[j][q]));

  // a bad association: add an inconsistent observation, optimise online, see chi2 jump, roll back
  obs.emplace_back();
  auto& bad = obs.back();
  bad.vertices()[0] = &poses[NP - 1];
  bad.vertices()[1] = &lines[0];
  bad.setMeasurement(Eigen::Vector2d(1.0, 2.5));
  Eigen::Matrix2d cov;
  cov << 0.004, 0, 0, 0.002;
  bad.information().noalias() = cov.inverse();
  opt.addEdge(&bad);
  HyperGraph::VertexSet nv;
  HyperGraph::EdgeSet ne;
  ne.insert(&bad);
  opt.updateInitialization(nv, ne);
  opt.push();
  opt.optimize(15, true);
  opt.computeActiveErrors();
  const double chi2_bad = opt.activeChi2();
  opt.removeEdge(&bad);
  opt.pop();                       // estimates back to the accepted state
  opt.initializeOptimization();
  opt.computeActiveErrors();
  const double chi2_restored = opt.activeChi2();

  std::cout << std::setprecision(12) << chi2_before << " " << chi2_after << " " << its << " " << perr << " " << lerr << " "
            << chi2_bad << " " << chi2_restored << " " << dof << std::endl;
  delete opt.algorithm();
  return 0;
}
