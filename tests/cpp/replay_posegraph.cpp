// replay_posegraph.cpp -- drives the g2o-compat shim (include/g2o/...) with the same g2o call
// sequences sparse-gslam uses on its pose graph, on a graph read from a text file:
//   solver stack        src/sparse_gslam/src/graphs.cpp:17-23 (setup_pose_opt), :32-37 (ctor/dtor)
//   fixed first pose    src/sparse_gslam/src/drone.cpp:70-76
//   chain + odometry    src/sparse_gslam/src/submap_loop_closer.cpp:208-224
//   closure + optimise  src/sparse_gslam/src/submap_loop_closer.cpp:272-288
//   final clean-up      src/sparse_gslam/src/log_runner.cpp:182-204 (chi2 gate 11.345, removeEdge)
// Written against the public API only; prints results for tests/test_shim_replay.py.
//
// usage: replay_posegraph graph.txt out.txt [gate]
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <unordered_set>

#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_gauss_newton.h"
#include "g2o/core/robust_kernel_impl.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/solvers/eigen/linear_solver_eigen.h"
#include "g2o/types/slam2d/edge_se2.h"
#include "g2o/types/slam2d/vertex_se2.h"

namespace {

struct PoseChain {
  g2o::VertexSE2 pose;
  g2o::EdgeSE2 edge;
};

struct PoseGraph {   // same ownership layout as graphs.h:31-40: deques own, optimiser borrows
  std::deque<PoseChain, Eigen::aligned_allocator<PoseChain>> poses;
  std::deque<g2o::EdgeSE2, Eigen::aligned_allocator<g2o::EdgeSE2>> all_closures;
  std::unordered_set<g2o::EdgeSE2*> closures, false_closures;
  g2o::SparseOptimizer opt;
  PoseGraph() {
    using SlamBlockSolver = g2o::BlockSolver<g2o::BlockSolverTraits<3, 3>>;
    using SlamLinearSolver = g2o::LinearSolverEigen<SlamBlockSolver::PoseMatrixType>;
    opt.setAlgorithm(new g2o::OptimizationAlgorithmGaussNewton(
        g2o::make_unique<SlamBlockSolver>(g2o::make_unique<SlamLinearSolver>())));
    opt.setVerbose(false);
    opt.setComputeBatchStatistics(false);
  }
  ~PoseGraph() { delete opt.algorithm(); }
};

g2o::RobustKernelDCS dcs_kernel;

Eigen::Matrix3d info_from(const double* u) {
  Eigen::Matrix3d O;
  O << u[0], u[1], u[2], u[1], u[3], u[4], u[2], u[4], u[5];
  return O;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const bool gate = argc > 3;
  std::ifstream in(argv[1]);
  int V, E;
  double phi;
  in >> V >> E >> phi;
  std::vector<double> poses(3 * (size_t)V);
  for (auto& v : poses) in >> v;
  struct Ed { int i, j; double z[3], o[6]; int closure; };
  std::vector<Ed> edges(E);
  for (auto& e : edges) {
    in >> e.i >> e.j >> e.closure;
    for (double& v : e.z) in >> v;
    for (double& v : e.o) in >> v;
  }
  if (!in) return 3;
  dcs_kernel.setDelta(phi);

  PoseGraph pg;
  // fixed first pose
  pg.poses.emplace_back();
  {
    auto* ip = &pg.poses.back().pose;
    ip->setId(0);
    ip->setEstimate(g2o::SE2(poses[0], poses[1], poses[2]));
    ip->setFixed(true);
    pg.opt.addVertex(ip);
  }
  // odometry chain (edge k-1 connects k-1 -> k; the file lists the V-1 odometry edges first)
  auto* prev_vertex = &pg.poses.back().pose;
  for (int k = 1; k < V; ++k) {
    const Ed& od = edges[k - 1];
    pg.poses.emplace_back();
    auto* pose = &pg.poses.back().pose;
    auto* edge = &pg.poses.back().edge;
    pose->setId(k);
    edge->vertices()[0] = prev_vertex;
    edge->vertices()[1] = pose;
    edge->information() = info_from(od.o);
    edge->setMeasurement(g2o::SE2(od.z[0], od.z[1], od.z[2]));
    pose->setEstimate(g2o::SE2(poses[3 * k], poses[3 * k + 1], poses[3 * k + 2]));
    pg.opt.addVertex(pose);
    pg.opt.addEdge(edge);
    prev_vertex = pose;
  }
  // loop closures with the shared DCS kernel
  for (int k = V - 1; k < E; ++k) {
    const Ed& c = edges[k];
    pg.all_closures.emplace_back();
    auto* ce = &pg.all_closures.back();
    ce->setMeasurement(g2o::SE2(c.z[0], c.z[1], c.z[2]));
    ce->information().noalias() = info_from(c.o).inverse().inverse();
    ce->vertices()[0] = pg.opt.vertices()[c.i];
    ce->vertices()[1] = &pg.poses[c.j].pose;
    ce->setRobustKernel(&dcs_kernel);
    pg.closures.insert(ce);
    pg.opt.addEdge(ce);
  }
  pg.opt.initializeOptimization();
  int it1 = pg.opt.optimize(20);
  pg.opt.computeActiveErrors();
  const double chi2_1 = pg.opt.activeChi2(), rchi2_1 = pg.opt.activeRobustChi2();

  int removed = 0, it2 = -2;
  double chi2_2 = 0, rchi2_2 = 0;
  if (gate) {
    for (auto& edge : pg.all_closures) {
      edge.computeError();
      if (edge.chi2() > 11.345) {
        pg.opt.removeEdge(&edge);
        pg.closures.erase(&edge);
        pg.false_closures.insert(&edge);
        ++removed;
      }
    }
    pg.opt.initializeOptimization();
    it2 = pg.opt.optimize(20);
    pg.opt.computeActiveErrors();
    chi2_2 = pg.opt.activeChi2();
    rchi2_2 = pg.opt.activeRobustChi2();
  }

  std::ofstream out(argv[2]);
  out << std::setprecision(17);
  out << it1 << " " << chi2_1 << " " << rchi2_1 << " " << removed << " " << it2 << " " << chi2_2 << " " << rchi2_2 << "\n";
  for (auto& pc : pg.poses) out << pc.pose.estimate()[0] << " " << pc.pose.estimate()[1] << " " << pc.pose.estimate()[2] << "\n";
  // a vertex looked up through the id map is the caller's object, and estimates are current
  auto* v5 = static_cast<g2o::VertexSE2*>(pg.opt.vertex(5));
  return (v5 == &pg.poses[5].pose && pg.opt.vertex(V + 7) == nullptr) ? 0 : 4;
}
