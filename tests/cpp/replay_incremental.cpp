// replay_incremental.cpp -- the loop closer's growth pattern through the g2o-compat shim: after every accepted closure
// the pose graph optimised before gets a chain of new poses with their odometry edges
// (src/sparse_gslam/src/submap_loop_closer.cpp:205-226: pose->setEstimate(prev->estimate() * edge->measurement())), one
// closure edge with the shared DCS kernel (:272-285), then initializeOptimization(); optimize(n) (:286-287).  The shim
// recognises the previous graph as a prefix of the new one and hands the backend an incremental update.
// Written against the public API only; prints results for tests/test_shim_replay.py.
//
// usage: replay_incremental session.txt out.txt iters
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>

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
struct Ed { int i, j; double z[3], o[6]; int closure; };
g2o::RobustKernelDCS dcs_kernel;
Eigen::Matrix3d info_from(const double* u) {
  Eigen::Matrix3d O;
  O << u[0], u[1], u[2], u[1], u[3], u[4], u[2], u[4], u[5];
  return O;
}
bool read_edges(std::istream& in, int n, std::vector<Ed>& out) {
  out.resize(n);
  for (auto& e : out) {
    in >> e.i >> e.j >> e.closure;
    for (double& v : e.z) in >> v;
    for (double& v : e.o) in >> v;
  }
  return (bool)in;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int iters = std::atoi(argv[3]);
  std::ifstream in(argv[1]);
  int V0, E0, nsteps;
  double phi;
  in >> V0 >> E0 >> nsteps >> phi;
  std::vector<double> poses(3 * (size_t)V0);
  for (auto& v : poses) in >> v;
  std::vector<Ed> edges;
  if (!read_edges(in, E0, edges)) return 3;
  dcs_kernel.setDelta(phi);

  std::deque<PoseChain, Eigen::aligned_allocator<PoseChain>> chain;
  std::deque<g2o::EdgeSE2, Eigen::aligned_allocator<g2o::EdgeSE2>> closures;
  g2o::SparseOptimizer opt;
  using SlamBlockSolver = g2o::BlockSolver<g2o::BlockSolverTraits<3, 3>>;
  using SlamLinearSolver = g2o::LinearSolverEigen<SlamBlockSolver::PoseMatrixType>;
  opt.setAlgorithm(new g2o::OptimizationAlgorithmGaussNewton(g2o::make_unique<SlamBlockSolver>(g2o::make_unique<SlamLinearSolver>())));

  auto add_closure = [&](const Ed& c) {
    closures.emplace_back();
    auto* ce = &closures.back();
    ce->setMeasurement(g2o::SE2(c.z[0], c.z[1], c.z[2]));
    ce->information() = info_from(c.o);
    ce->vertices()[0] = opt.vertices()[c.i];
    ce->vertices()[1] = &chain[c.j].pose;
    ce->setRobustKernel(&dcs_kernel);
    opt.addEdge(ce);
  };
  // the graph optimised so far: fixed first pose, odometry chain (the file lists the V0 - 1 odometry edges first), closures
  chain.emplace_back();
  chain.back().pose.setId(0);
  chain.back().pose.setEstimate(g2o::SE2(poses[0], poses[1], poses[2]));
  chain.back().pose.setFixed(true);
  opt.addVertex(&chain.back().pose);
  for (int k = 1; k < V0; ++k) {
    const Ed& od = edges[k - 1];
    auto* prev = &chain.back().pose;
    chain.emplace_back();
    auto* pose = &chain.back().pose;
    auto* edge = &chain.back().edge;
    pose->setId(k);
    edge->vertices()[0] = prev;
    edge->vertices()[1] = pose;
    edge->information() = info_from(od.o);
    edge->setMeasurement(g2o::SE2(od.z[0], od.z[1], od.z[2]));
    pose->setEstimate(g2o::SE2(poses[3 * k], poses[3 * k + 1], poses[3 * k + 2]));
    opt.addVertex(pose);
    opt.addEdge(edge);
  }
  for (int k = V0 - 1; k < E0; ++k) add_closure(edges[k]);
  std::ofstream out(argv[2]);
  out << std::setprecision(17);
  opt.initializeOptimization();
  int done = opt.optimize(iters);
  opt.computeActiveErrors();
  out << done << " " << opt.activeChi2() << " " << opt.activeRobustChi2() << " | " << opt.backendDescription() << "\n";
  for (int s = 0; s < nsteps; ++s) {
    int Vs, nE;
    in >> Vs >> nE;
    std::vector<Ed> app;
    if (!read_edges(in, nE, app)) return 3;
    for (const Ed& e : app) {
      if (e.closure) continue;
      // slc.cpp:208-224: the odometry edge (prev -> new pose), the new pose chained from prev's CURRENT estimate
      auto* prev = &chain.back().pose;
      if (prev->id() != e.i || e.j != e.i + 1) return 5;
      chain.emplace_back();
      auto* pose = &chain.back().pose;
      auto* edge = &chain.back().edge;
      pose->setId(e.j);
      edge->vertices()[0] = prev;
      edge->vertices()[1] = pose;
      edge->information() = info_from(e.o);
      edge->setMeasurement(g2o::SE2(e.z[0], e.z[1], e.z[2]));
      pose->setEstimate(prev->estimate() * edge->measurement());
      opt.addVertex(pose);
      opt.addEdge(edge);
    }
    for (const Ed& e : app)
      if (e.closure) add_closure(e);
    if ((int)chain.size() != Vs) return 6;
    opt.initializeOptimization();
    done = opt.optimize(iters);
    opt.computeActiveErrors();
    out << done << " " << opt.activeChi2() << " " << opt.activeRobustChi2() << " | " << opt.backendDescription() << "\n";
  }
  for (auto& pc : chain) out << pc.pose.estimate()[0] << " " << pc.pose.estimate()[1] << " " << pc.pose.estimate()[2] << "\n";
  delete opt.algorithm();
  return 0;
}
