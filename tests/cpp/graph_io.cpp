// graph_io.cpp -- g2o text files through the compat header: SparseOptimizer::load / save (VERTEX_SE2, EDGE_SE2,
// FIX; EdgeSE2::read / write), the CARMEN result line of src/sparse_gslam/src/log_runner.cpp:18-23, and -- with
// "optimize" as third argument, on a GPU -- optimize(20) of the loaded graph (the pose-graph solver stack of
// src/sparse_gslam/src/graphs.cpp:17-23).
// usage: graph_io <in.g2o> <out.g2o> [optimize <result.carmen>]
#include <algorithm>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_gauss_newton.h"
#include "g2o/core/robust_kernel_impl.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/solvers/eigen/linear_solver_eigen.h"
#include "g2o/types/slam2d/edge_se2.h"
#include "g2o/types/slam2d/vertex_se2.h"

int main(int argc, char** argv) {
  using namespace g2o;
  if (argc < 3) return 2;
  using SlamBlockSolver = BlockSolver<BlockSolverTraits<3, 3>>;
  using SlamLinearSolver = LinearSolverEigen<SlamBlockSolver::PoseMatrixType>;
  SparseOptimizer opt;
  opt.setAlgorithm(new OptimizationAlgorithmGaussNewton(g2o::make_unique<SlamBlockSolver>(g2o::make_unique<SlamLinearSolver>())));
  if (!opt.load(argv[1])) {
    std::cerr << "load failed" << std::endl;
    return 1;
  }
  int nfixed = 0;
  for (auto& kv : opt.vertices()) nfixed += static_cast<OptimizableGraph::Vertex*>(kv.second)->fixed();
  std::cout << std::setprecision(17) << opt.vertices().size() << " " << opt.edges().size() << " " << nfixed << std::endl;
  if (argc >= 5 && std::string(argv[3]) == "optimize") {
    // loop closures (non-consecutive ids) carry the shared DCS kernel, as submap_loop_closer.cpp:283
    RobustKernelDCS dcs;
    dcs.setDelta(1.0);
    for (auto* he : opt.edges()) {
      auto* e = static_cast<OptimizableGraph::Edge*>(he);
      if (std::abs(e->vertices()[0]->id() - e->vertices()[1]->id()) != 1) e->setRobustKernel(&dcs);
    }
    opt.initializeOptimization();
    const int its = opt.optimize(20);
    opt.computeActiveErrors();
    std::cout << its << " " << opt.activeChi2() << " " << opt.activeRobustChi2() << std::endl;
    std::ofstream res(argv[4]);
    double t = 0.0;
    std::vector<int> ids;   // vertices() is a hash map: the trajectory file goes in id (= time) order
    for (auto& kv : opt.vertices()) ids.push_back(kv.first);
    std::sort(ids.begin(), ids.end());
    for (int id : ids) write_carmen_result_line(res, static_cast<VertexSE2*>(opt.vertex(id))->estimate(), t++);
  }
  if (!opt.save(argv[2])) return 1;
  delete opt.algorithm();
  return 0;
}
