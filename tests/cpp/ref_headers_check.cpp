// ref_headers_check.cpp -- compiles the REFERENCE's own custom-type declarations
// (src/sparse_gslam/include/g2o_bindings/{vertex_rhotheta,edge_se2_rhotheta}.h, included from the
// read-only reference checkout, never copied) against the g2o compat headers of this repo.  The
// member definitions below are this test's own (the reference's .cpp files additionally need
// ls_extractor, Boost and real Eigen); what is checked is that the subclass ABI the reference
// declares -- BaseVertex<2, Eigen::Vector2d>, BaseBinaryEdge<2, Eigen::Vector2d, VertexSE2,
// VertexRhoTheta>, the overridden virtuals, EIGEN_MAKE_ALIGNED_OPERATOR_NEW, G2O_REGISTER_TYPE --
// is accepted, and that such objects go through a Levenberg-configured SparseOptimizer.
#include <iostream>

#include "g2o/core/block_solver.h"
#include "g2o/core/factory.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/solvers/eigen/linear_solver_eigen.h"
#include "g2o/stuff/macros.h"
#include "delta_vector.h"   // reference header: struct Delta { double dt; g2o::SE2 dpose; } in an aligned vector
#include "g2o_bindings/edge_se2_rhotheta.h"
#include "g2o_bindings/vertex_rhotheta.h"

namespace g2o {
void VertexRhoTheta::updateEndpoints() {}
void VertexRhoTheta::setToOriginImpl() { _estimate.setZero(); }
void VertexRhoTheta::oplusImpl(const double* update) {
  _estimate[0] += update[0];
  _estimate[1] += update[1];
}
bool VertexRhoTheta::read(std::istream&) { return true; }
bool VertexRhoTheta::write(std::ostream& os) const { return os.good(); }
void EdgeSE2RhoTheta::computeError() {
  const auto* pose = static_cast<VertexSE2*>(_vertices[0]);
  const auto* line = static_cast<VertexRhoTheta*>(_vertices[1]);
  const double th = line->estimate()[1] - pose->estimate().rotation().angle();
  const double rho = line->estimate()[0] - (pose->estimate().translation()[0] * std::cos(line->estimate()[1]) +
                                            pose->estimate().translation()[1] * std::sin(line->estimate()[1]));
  _error[0] = _measurement[0] - rho;
  _error[1] = normalize_theta(_measurement[1] - th);
}
bool EdgeSE2RhoTheta::read(std::istream&) { return true; }
bool EdgeSE2RhoTheta::write(std::ostream&) const { return true; }
G2O_REGISTER_TYPE(VERTEX_RHOTHETA, VertexRhoTheta);
G2O_REGISTER_TYPE(EDGE_SE2_RHOTHETA, EdgeSE2RhoTheta);
}  // namespace g2o

int main() {
  using namespace g2o;
  using SlamBlockSolver = BlockSolver<BlockSolverTraits<-1, 2>>;
  using SlamLinearSolver = LinearSolverEigen<SlamBlockSolver::PoseMatrixType>;
  SparseOptimizer opt;
  opt.setAlgorithm(new OptimizationAlgorithmLevenberg(g2o::make_unique<SlamBlockSolver>(g2o::make_unique<SlamLinearSolver>())));
  VertexSE2 p0, p1;
  VertexRhoTheta lm;
  EdgeSE2 od;
  EdgeSE2RhoTheta ob0, ob1;
  p0.setId(0); p0.setEstimate(SE2(0, 0, 0)); p0.setFixed(true);
  p1.setId(1); p1.setEstimate(SE2(0.9, 0.1, 0.05));
  lm.setId(10000000); lm.setEstimate(Eigen::Vector2d(2.1, 0.05));
  lm.start = Eigen::Vector2f(0, 0); lm.end = Eigen::Vector2f(1, 1); lm.dist = 0;
  od.vertices()[0] = &p0; od.vertices()[1] = &p1; od.setMeasurement(SE2(1, 0, 0));
  ob0.vertices()[0] = &p0; ob0.vertices()[1] = &lm; ob0.setMeasurement(Eigen::Vector2d(2.0, 0.0));
  ob1.vertices()[0] = &p1; ob1.vertices()[1] = &lm; ob1.setMeasurement(Eigen::Vector2d(1.0, 0.0));
  Eigen::Matrix2d cov;
  cov << 0.01, 0, 0, 0.01;
  ob0.information().noalias() = cov.inverse();
  ob1.information().noalias() = cov.inverse();
  opt.addVertex(&p0); opt.addVertex(&p1); opt.addVertex(&lm);
  opt.addEdge(&od); opt.addEdge(&ob0); opt.addEdge(&ob1);
  opt.initializeOptimization();
  opt.push();
  const int its = opt.optimize(15, false);
  opt.computeActiveErrors();
  const double chi2 = opt.activeChi2();
  opt.discardTop();
  DeltaVector dv;
  dv.push_back({0.1, p1.estimate()});
  if (dv.size() != 1 || dv[0].dpose[0] != p1.estimate()[0]) return 2;
  std::cout << its << " " << chi2 << " " << lm.estimate()[0] << " " << p1.estimate()[0] << std::endl;
  delete opt.algorithm();
  return (its >= 1 && chi2 < 1e-12) ? 0 : 1;
}
