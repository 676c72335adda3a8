// landmark_rhotheta.cpp -- the landmark graph with the REFERENCE's edge model, against the numpy LM oracle's
// golden vectors (tests/golden/lm_landmark.json, scripts/make_golden_lm.py).
//
// The vertex / edge classes restate include/g2o_bindings/{vertex_rhotheta,edge_se2_rhotheta}.h and
// src/sparse_gslam/src/g2o_bindings/{vertex_rhotheta.cpp:28-34, edge_se2_rhotheta.cpp:9-16} in this test's own
// words (the reference's files themselves are compiled against the compat headers by ref_headers_check.cpp,
// where the checkout is present): a 2-dof (rho, theta) vertex with an additive oplus whose theta is NOT wrapped
// (upstream discards normalize_theta's result), and a computeError-only edge
//     e = z - transform_line(line, pose^-1),  e[1] wrapped,
// with transform_line / checkRhoTheta restated from src/ls_extractor/include/ls_extractor/utils.h:22-45.
// Call sequence as src/sparse_gslam/src/drone.cpp:146-156: initializeOptimization / push / optimize(15, false),
// then new vertices + edges / updateInitialization / push / optimize(15, true).
//
// With -DSGO_REF_SOURCES the restated classes are left out and the program uses the REFERENCE's own g2o::VertexRhoTheta /
// g2o::EdgeSE2RhoTheta: declarations from include/g2o_bindings/*.h, definitions from src/g2o_bindings/{vertex_rhotheta,
// edge_se2_rhotheta}.cpp compiled verbatim from the read-only checkout and linked in (tests/test_landmark_lm.py, CPU box
// only) -- the same schedule then runs through the reference's computeError / oplusImpl and ls_extractor/utils.h.
//
// usage: landmark_rhotheta <graph.txt>; prints, per stage, "STAGE iterations chi2" and one
// "IT lambda chi2 trials" line per iteration, then "V id est...".
#include <cmath>
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <string>

#include "g2o/core/base_binary_edge.h"
#include "g2o/core/base_vertex.h"
#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/solvers/eigen/linear_solver_eigen.h"
#include "g2o/stuff/misc.h"
#include "g2o/types/slam2d/edge_se2.h"
#include "g2o/types/slam2d/vertex_se2.h"

#ifdef SGO_REF_SOURCES
#include "g2o_bindings/edge_se2_rhotheta.h"
#include "g2o_bindings/vertex_rhotheta.h"
namespace g2o {
using VertexLineRT = VertexRhoTheta;
using EdgePoseLineRT = EdgeSE2RhoTheta;
}  // namespace g2o
#else
namespace g2o {

inline Eigen::Vector2d move_line(const Eigen::Vector2d& rt, const Eigen::Vector2d& trans, double angle) {
  double th = rt[1] + angle;
  if (th > const_pi()) th -= 2 * const_pi();
  if (th < -const_pi()) th += 2 * const_pi();
  double rho = rt[0] + trans[0] * std::cos(th) + trans[1] * std::sin(th);
  if (rho < 0.0) {
    rho = -rho;
    th += const_pi();
    if (th > const_pi()) th -= 2 * const_pi();
  }
  return Eigen::Vector2d(rho, th);
}

class VertexLineRT : public BaseVertex<2, Eigen::Vector2d> {
 public:
  void setToOriginImpl() override { _estimate.setZero(); }
  void oplusImpl(const double* u) override {
    _estimate[0] += u[0];
    _estimate[1] += u[1];
    normalize_theta(_estimate[1]);   // result discarded, as upstream
  }
  bool read(std::istream&) override { return true; }
  bool write(std::ostream& os) const override { return os.good(); }
};

class EdgePoseLineRT : public BaseBinaryEdge<2, Eigen::Vector2d, VertexSE2, VertexLineRT> {
 public:
  void computeError() override {
    const auto* pose = static_cast<VertexSE2*>(_vertices[0]);
    const auto* line = static_cast<VertexLineRT*>(_vertices[1]);
    const SE2 pinv = pose->estimate().inverse();
    const Eigen::Vector2d pred = move_line(line->estimate(), pinv.translation(), pinv.rotation().angle());
    _error[0] = _measurement[0] - pred[0];
    _error[1] = normalize_theta(_measurement[1] - pred[1]);
  }
  bool read(std::istream&) override { return true; }
  bool write(std::ostream& os) const override { return os.good(); }
};

}  // namespace g2o
#endif

int main(int argc, char** argv) {
  using namespace g2o;
  if (argc < 2) return 2;
  using SlamBlockSolver = BlockSolver<BlockSolverTraits<-1, 2>>;
  using SlamLinearSolver = LinearSolverEigen<SlamBlockSolver::PoseMatrixType>;
  SparseOptimizer opt;
  opt.setAlgorithm(new OptimizationAlgorithmLevenberg(g2o::make_unique<SlamBlockSolver>(g2o::make_unique<SlamLinearSolver>())));
  std::deque<VertexSE2> poses;
  std::deque<VertexLineRT> lines;
  std::deque<EdgeSE2> odom;
  std::deque<EdgePoseLineRT> obs;
  std::map<int, OptimizableGraph::Vertex*> byid;
  std::cout << std::setprecision(17);
  for (int stage = 0; stage < 2; ++stage) {
    HyperGraph::VertexSet nv;
    HyperGraph::EdgeSet ne;
    std::ifstream in(argv[1]);
    std::string kind;
    int st;
    while (in >> kind >> st) {
      if (kind == "POSE") {
        int id, fixed;
        double x, y, t;
        in >> id >> x >> y >> t >> fixed;
        if (st != stage) continue;
        poses.emplace_back();
        poses.back().setId(id);
        poses.back().setEstimate(SE2(x, y, t));
        poses.back().setFixed(fixed != 0);
        opt.addVertex(&poses.back());
        byid[id] = &poses.back();
        nv.insert(&poses.back());
      } else if (kind == "LINE") {
        int id;
        double r, t;
        in >> id >> r >> t;
        if (st != stage) continue;
        lines.emplace_back();
        lines.back().setId(id);
        lines.back().setEstimate(Eigen::Vector2d(r, t));
        opt.addVertex(&lines.back());
        byid[id] = &lines.back();
        nv.insert(&lines.back());
      } else if (kind == "ODOM") {
        int i, j;
        double z[3], u[6];
        in >> i >> j >> z[0] >> z[1] >> z[2];
        for (double& v : u) in >> v;
        if (st != stage) continue;
        odom.emplace_back();
        auto& e = odom.back();
        e.vertices()[0] = byid[i];
        e.vertices()[1] = byid[j];
        e.setMeasurement(SE2(z[0], z[1], z[2]));
        Eigen::Matrix3d O;
        O << u[0], u[1], u[2], u[1], u[3], u[4], u[2], u[4], u[5];
        e.information() = O;
        opt.addEdge(&e);
        ne.insert(&e);
      } else if (kind == "OBS") {
        int i, j;
        double z[2], u[3];
        in >> i >> j >> z[0] >> z[1] >> u[0] >> u[1] >> u[2];
        if (st != stage) continue;
        obs.emplace_back();
        auto& e = obs.back();
        e.vertices()[0] = byid[i];
        e.vertices()[1] = byid[j];
        e.setMeasurement(Eigen::Vector2d(z[0], z[1]));
        Eigen::Matrix2d O;
        O << u[0], u[1], u[1], u[2];
        e.information() = O;
        opt.addEdge(&e);
        ne.insert(&e);
      }
    }
    if (stage == 0) opt.initializeOptimization();
    else opt.updateInitialization(nv, ne);
    opt.push();
    const int its = opt.optimize(15, stage == 1);
    opt.computeActiveErrors();
    opt.discardTop();
    std::cout << "STAGE " << its << " " << opt.activeChi2() << "\n";
    for (const auto& t : opt.lmTrace()) std::cout << "IT " << t.lambda << " " << t.chi2 << " " << t.trials << "\n";
  }
  for (auto& kv : byid) {
    std::cout << "V " << kv.first;
    if (auto* p = dynamic_cast<VertexSE2*>(kv.second)) std::cout << " " << p->estimate()[0] << " " << p->estimate()[1] << " " << p->estimate()[2];
    else {
      auto* l = static_cast<VertexLineRT*>(kv.second);
      std::cout << " " << l->estimate()[0] << " " << l->estimate()[1];
    }
    std::cout << "\n";
  }
  delete opt.algorithm();
  return 0;
}
