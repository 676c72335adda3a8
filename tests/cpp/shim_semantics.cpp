// shim_semantics.cpp -- container / bookkeeping semantics of the g2o-compat SparseOptimizer that the
// reference relies on (SURVEY.md section 8(b) "semantic notes"); host only, no GPU call is made.
// Prints "ok" and returns 0 when every check holds.
#include <deque>
#include <iostream>

#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_gauss_newton.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/solvers/eigen/linear_solver_eigen.h"
#include "g2o/types/slam2d/edge_se2.h"
#include "g2o/types/slam2d/vertex_se2.h"

#define CHECK(c)                                                            \
  do {                                                                      \
    if (!(c)) {                                                             \
      std::cerr << "FAILED line " << __LINE__ << ": " #c << std::endl;      \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main() {
  using namespace g2o;
  VertexSE2 v[4];
  EdgeSE2 e01, e12, e23;
  for (int k = 0; k < 4; ++k) {
    v[k].setId(10 * (3 - k));   // ids 30, 20, 10, 0: hessian order must follow ids, not insertion
    v[k].setEstimate(SE2(k, 0, 0));
  }
  v[3].setFixed(true);          // id 0
  {
    SparseOptimizer opt;
    CHECK(opt.optimize(3) == -1);                       // no algorithm: nothing happens
    for (int k = 0; k < 4; ++k) CHECK(opt.addVertex(&v[k]));
    CHECK(!opt.addVertex(&v[1]));                       // duplicate id
    VertexSE2 dup;
    dup.setId(20);
    CHECK(!opt.addVertex(&dup));
    e01.vertices()[0] = &v[0]; e01.vertices()[1] = &v[1];
    e12.vertices()[0] = &v[1]; e12.vertices()[1] = &v[2];
    e23.vertices()[0] = &v[2]; e23.vertices()[1] = &v[3];
    for (EdgeSE2* e : {&e01, &e12, &e23}) e->setMeasurement(SE2(1, 0, 0));
    CHECK(opt.addEdge(&e01) && opt.addEdge(&e12) && opt.addEdge(&e23));
    CHECK(!opt.addEdge(&e12));                          // already in the graph
    CHECK(opt.vertex(10) == &v[2] && opt.vertex(11) == nullptr);
    CHECK(opt.vertices()[30] == &v[0]);
    CHECK(opt.initializeOptimization());
    CHECK(opt.activeVertices().size() == 4 && opt.activeEdges().size() == 3);
    CHECK(opt.activeVertices()[0] == &v[3] && opt.activeVertices()[3] == &v[0]);   // ascending id
    CHECK(v[3].hessianIndex() == -1 && v[2].hessianIndex() == 0 && v[1].hessianIndex() == 1 && v[0].hessianIndex() == 2);
    CHECK(opt.activeEdges()[0] == &e01 && opt.activeEdges()[2] == &e23);           // insertion order
    opt.computeActiveErrors();
    CHECK(opt.activeChi2() > 0 || opt.activeChi2() == 0);

    // removeEdge detaches the edge from its vertices
    CHECK(opt.removeEdge(&e12) && !opt.removeEdge(&e12));
    CHECK(v[1].edges().count(&e12) == 0 && v[2].edges().count(&e12) == 0);
    opt.initializeOptimization();
    CHECK(opt.activeEdges().size() == 2);
    CHECK(opt.addEdge(&e12));

    // clear(): containers emptied, the vertices' own edge sets are not (upstream semantics);
    // a re-added vertex alone activates nothing, re-adding its neighbour revives the old edge
    opt.clear();
    CHECK(opt.vertices().empty() && opt.edges().empty());
    CHECK(v[0].edges().size() == 1 && v[1].edges().size() == 2);
    CHECK(opt.addVertex(&v[0]));
    opt.initializeOptimization();
    CHECK(opt.activeEdges().empty() && opt.activeVertices().empty());
    CHECK(opt.addVertex(&v[1]));
    opt.initializeOptimization();
    CHECK(opt.activeEdges().size() == 1 && opt.activeEdges()[0] == &e01 && opt.activeVertices().size() == 2);

    // removeVertex only detaches edges that are still in the graph's own container: the stale
    // edge survives in the neighbour's set, exactly as upstream (removeEdge fails for it)
    CHECK(opt.removeVertex(&v[1]));
    CHECK(v[0].edges().size() == 1 && opt.vertex(20) == nullptr);
    // push / pop / discardTop act on the vertices' estimate stacks
    opt.addVertex(&v[1]);
    e01.vertices()[0] = &v[0]; e01.vertices()[1] = &v[1];
    opt.addEdge(&e01);
    opt.initializeOptimization();
    opt.push();
    v[0].setEstimate(SE2(9, 9, 1));
    opt.pop();
    CHECK(v[0].estimate()[0] == 0.0 && v[0].stackSize() == 0);
    opt.push();
    opt.discardTop();
    CHECK(v[0].stackSize() == 0);
  }  // ~SparseOptimizer must not delete or touch the caller's vertices / edges
  CHECK(v[0].id() == 30 && e23.vertices()[1] == &v[3]);
  // the robust kernel maths
  RobustKernelDCS k;
  k.setDelta(1.0);
  Vector3 rho;
  k.robustify(3.0, rho);
  CHECK(rho[0] == 0.75 && rho[1] == 0.25 && rho[2] == 0.0);
  k.robustify(0.5, rho);
  CHECK(rho[0] == 0.5 && rho[1] == 1.0);
  CHECK(normalize_theta(3 * const_pi()) == -const_pi());
  {
    // A large pose graph in a configuration the device path does not cover (Levenberg-Marquardt)
    // is refused (optimize() == 0, estimates untouched): the dense host solver serves the
    // landmark graph only and must never become a silent fallback for pose graphs.
    const int N = 1200;
    std::deque<VertexSE2> pv(N);
    std::deque<EdgeSE2> pe(N - 1);
    SparseOptimizer big;
    auto* alg = new OptimizationAlgorithmLevenberg(
        g2o::make_unique<BlockSolver<BlockSolverTraits<3, 3>>>(
            g2o::make_unique<LinearSolverEigen<BlockSolver<BlockSolverTraits<3, 3>>::PoseMatrixType>>()));
    big.setAlgorithm(alg);
    for (int k = 0; k < N; ++k) {
      pv[k].setId(k);
      pv[k].setEstimate(SE2(k, 0, 0));
      pv[k].setFixed(k == 0);
      big.addVertex(&pv[k]);
    }
    for (int k = 0; k + 1 < N; ++k) {
      pe[k].vertices()[0] = &pv[k];
      pe[k].vertices()[1] = &pv[k + 1];
      pe[k].setMeasurement(SE2(1.1, 0, 0));
      pe[k].information().setIdentity();
      big.addEdge(&pe[k]);
    }
    big.initializeOptimization();
    CHECK(big.optimize(5) == 0);
    CHECK(pv[N - 1].estimate()[0] == double(N - 1));
    delete alg;
  }
  std::cout << "ok" << std::endl;
  return 0;
}
