// sgo_g2o_compat.h -- header-only C++ mirror of the g2o API surface that sparse-gslam uses,
// implemented over the C-ABI of libsgo (include/sgo.h).  The individual g2o header paths
// (g2o/core/sparse_optimizer.h, g2o/types/slam2d/edge_se2.h, ...) include this file, so the
// reference's sources keep their #include lines (src/sparse_gslam/include/graphs.h:2-4,
// src/sparse_gslam/src/graphs.cpp:3-7, src/sparse_gslam/src/submap_loop_closer.cpp:8-9,
// src/sparse_gslam/include/delta_vector.h:2, include/g2o_bindings/*.h).
//
// Semantics mirrored (upstream g2o 2020.5.29):
//   * the caller owns vertices, edges, kernels and the algorithm object; nothing is deleted here
//     (README.md:22-23 G2O_DELETE_IMPLICITLY_OWNED_OBJECTS 0; graphs.cpp:29,36 delete the
//     algorithm before ~SparseOptimizer runs);
//   * add*/remove* return bool, optimize() returns the iteration count (0: solver failed,
//     -1: nothing to optimise); diagnostics go to std::cerr; no exceptions;
//   * initializeOptimization(): active vertices = graph vertices with at least one edge whose
//     vertices are all in the graph, sorted by id; active edges in insertion order; hessian
//     indices = non-fixed active vertices in ascending id;
//   * HyperGraph::clear() empties the graph's containers but not each vertex's own edges() set;
//   * activeChi2() is the un-robustified sum, activeRobustChi2() the robustified one.
//
// Backend: a graph whose active vertices are all VertexSE2 and whose active edges are all EdgeSE2
// (robust kernel: none or RobustKernelDCS) optimised with OptimizationAlgorithmGaussNewton -- the
// reference's pose graph (graphs.cpp:17-23), the hot path -- ALWAYS runs on the GPU through
// sgo_optimize_gn; if the device or libsgo is unavailable optimize() fails loudly, there is no
// CPU fallback for it.  Every other combination (the reference's landmark graph: Levenberg,
// BlockSolverTraits<-1,2>, VertexRhoTheta / EdgeSE2RhoTheta with numeric Jacobians, graphs.cpp:9-15,
// drone.cpp:146-187) is a tiny, latency-bound problem (pruned to one pose at every loop closure,
// slc.cpp:257-262) and is solved by the small dense host solver at the end of this file
// ("CPU path of the new backend suffices").
#pragma once

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(SGO_FORCE_EIGEN_MIN)
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/StdVector>
#else
#include "sgo_eigen_min.h"
#endif
#else
#include "sgo_eigen_min.h"
#endif

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <typeinfo>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../sgo.h"

#ifndef G2O_REGISTER_TYPE
#define G2O_REGISTER_TYPE(name, classname)
#endif
#ifndef G2O_ATTRIBUTE_UNUSED
#define G2O_ATTRIBUTE_UNUSED __attribute__((unused))
#endif

namespace g2o {

using number_t = double;
using Vector2 = Eigen::Matrix<double, 2, 1>;
using Vector3 = Eigen::Matrix<double, 3, 1>;
using Matrix3 = Eigen::Matrix<double, 3, 3>;
using Rotation2D = Eigen::Rotation2D<double>;

template <class T, class... A>
std::unique_ptr<T> make_unique(A&&... a) {
  return std::unique_ptr<T>(new T(std::forward<A>(a)...));
}

inline constexpr double const_pi() { return 3.14159265358979323846; }

// g2o/stuff/misc.h
inline double normalize_theta(double theta) {
  if (theta >= -const_pi() && theta < const_pi()) return theta;
  double multiplier = std::floor(theta / (2 * const_pi()));
  theta = theta - multiplier * 2 * const_pi();
  if (theta >= const_pi()) theta -= 2 * const_pi();
  if (theta < -const_pi()) theta += 2 * const_pi();
  return theta;
}

// ------------------------------------------------------------------------------- SE2
class SE2 {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  SE2() : _R(0), _t(0, 0) {}
  SE2(double x, double y, double theta) : _R(theta), _t(x, y) {}
  const Vector2& translation() const { return _t; }
  void setTranslation(const Vector2& t) { _t = t; }
  const Rotation2D& rotation() const { return _R; }
  void setRotation(const Rotation2D& R) { _R = R; }

  SE2 operator*(const SE2& tr2) const {
    SE2 result(*this);
    result *= tr2;
    return result;
  }
  SE2& operator*=(const SE2& tr2) {
    _t = _t + _R * tr2._t;
    _R.angle() += tr2._R.angle();
    _R.angle() = normalize_theta(_R.angle());
    return *this;
  }
  Vector2 operator*(const Vector2& v) const { return _t + _R * v; }
  SE2 inverse() const {
    SE2 ret;
    ret._R = _R.inverse();
    ret._R.angle() = normalize_theta(ret._R.angle());
    ret._t = ret._R * (Vector2(-_t[0], -_t[1]));
    return ret;
  }
  double operator[](int i) const {
    assert(i >= 0 && i < 3);
    if (i < 2) return _t[i];
    return _R.angle();
  }
  void fromVector(const Vector3& v) { *this = SE2(v[0], v[1], v[2]); }
  Vector3 toVector() const { return Vector3(_t[0], _t[1], _R.angle()); }

 protected:
  Rotation2D _R;
  Vector2 _t;
};

// ------------------------------------------------------------------------------- graph
class SparseOptimizer;
class RobustKernel;

class HyperGraph {
 public:
  class Edge;
  class Vertex {
   public:
    explicit Vertex(int id = -1) : _id(id) {}
    virtual ~Vertex() {}
    int id() const { return _id; }
    virtual void setId(int id) { _id = id; }
    const std::set<Edge*>& edges() const { return _edges; }
    std::set<Edge*>& edges() { return _edges; }

   protected:
    int _id;
    std::set<Edge*> _edges;
  };
  class Edge {
   public:
    explicit Edge(int id = -1) : _id(id) {}
    virtual ~Edge() {}
    const std::vector<Vertex*>& vertices() const { return _vertices; }
    std::vector<Vertex*>& vertices() { return _vertices; }
    const Vertex* vertex(size_t i) const { return _vertices[i]; }
    Vertex* vertex(size_t i) { return _vertices[i]; }
    void setVertex(size_t i, Vertex* v) { _vertices[i] = v; }
    int id() const { return _id; }
    void setId(int id) { _id = id; }
    long long internalId() const { return _internalId; }

   protected:
    friend class HyperGraph;
    friend class SparseOptimizer;
    std::vector<Vertex*> _vertices;
    int _id;
    long long _internalId = -1;
  };
  using VertexSet = std::set<Vertex*>;
  using EdgeSet = std::set<Edge*>;
  using VertexIDMap = std::unordered_map<int, Vertex*>;
  using VertexContainer = std::vector<Vertex*>;
  using EdgeContainer = std::vector<Edge*>;
};

class OptimizableGraph : public HyperGraph {
 public:
  class Vertex : public HyperGraph::Vertex {
   public:
    virtual ~Vertex() {}
    bool fixed() const { return _fixed; }
    void setFixed(bool f) { _fixed = f; }
    int hessianIndex() const { return _hessianIndex; }
    void setHessianIndex(int i) { _hessianIndex = i; }
    int colInHessian() const { return _colInHessian; }
    void setColInHessian(int c) { _colInHessian = c; }
    int tempIndex() const { return _tempIndex; }
    void setTempIndex(int t) { _tempIndex = t; }
    virtual int dimension() const = 0;
    virtual void oplus(const double* v) = 0;
    virtual void push() = 0;
    virtual void pop() = 0;
    virtual void discardTop() = 0;
    virtual void setToOrigin() = 0;

   protected:
    bool _fixed = false;
    int _hessianIndex = -1;
    int _colInHessian = -1;
    int _tempIndex = -1;
  };
  class Edge : public HyperGraph::Edge {
   public:
    virtual ~Edge() {}
    virtual int dimension() const = 0;
    virtual void computeError() = 0;
    virtual double chi2() const = 0;
    virtual void linearizeOplus() = 0;
    // adds this edge's J^T W J / -J^T W e into the dense normal equations (row-major H, ld = n)
    virtual void constructQuadraticForm(double* H, int n, double* b) = 0;
    RobustKernel* robustKernel() const { return _robustKernel; }
    void setRobustKernel(RobustKernel* k) { _robustKernel = k; }
    int level() const { return _level; }
    void setLevel(int l) { _level = l; }

   protected:
    RobustKernel* _robustKernel = nullptr;
    int _level = 0;
  };
};

// g2o/core/robust_kernel.h / robust_kernel_impl.h
class RobustKernel {
 public:
  RobustKernel() : _delta(1.) {}
  virtual ~RobustKernel() {}
  virtual void robustify(double squaredError, Vector3& rho) const = 0;
  virtual void setDelta(double delta) { _delta = delta; }
  double delta() const { return _delta; }

 protected:
  double _delta;
};
class RobustKernelDCS : public RobustKernel {
 public:
  void robustify(double e2, Vector3& rho) const override {
    const double& phi = _delta;
    double scale = (2.0 * phi) / (phi + e2);
    if (scale >= 1.0) {
      rho[0] = e2;
      rho[1] = 1.;
      rho[2] = 0;
    } else {
      rho[0] = scale * e2 * scale;
      rho[1] = (scale * scale);
      rho[2] = 0;
    }
  }
};

// g2o/core/base_vertex.h
template <int D, typename T>
class BaseVertex : public OptimizableGraph::Vertex {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  using EstimateType = T;
  static const int Dimension = D;
  int dimension() const override { return D; }
  const EstimateType& estimate() const { return _estimate; }
  void setEstimate(const EstimateType& et) { _estimate = et; }
  void oplus(const double* v) override { oplusImpl(v); }
  void setToOrigin() override { setToOriginImpl(); }
  void push() override { _backup.push_back(_estimate); }
  void pop() override {
    assert(!_backup.empty());
    _estimate = _backup.back();
    _backup.pop_back();
  }
  void discardTop() override {
    assert(!_backup.empty());
    _backup.pop_back();
  }
  int stackSize() const { return (int)_backup.size(); }
  virtual void setToOriginImpl() = 0;
  virtual void oplusImpl(const double* v) = 0;
  virtual bool read(std::istream& is) = 0;
  virtual bool write(std::ostream& os) const = 0;

 protected:
  EstimateType _estimate;
  std::vector<EstimateType, Eigen::aligned_allocator<EstimateType>> _backup;
};

// g2o/core/base_binary_edge.h
template <int D, typename E, typename VertexXi, typename VertexXj>
class BaseBinaryEdge : public OptimizableGraph::Edge {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  using Measurement = E;
  using VertexXiType = VertexXi;
  using VertexXjType = VertexXj;
  using ErrorVector = Eigen::Matrix<double, D, 1>;
  using InformationType = Eigen::Matrix<double, D, D>;
  using JacobianXiOplusType = Eigen::Matrix<double, D, VertexXi::Dimension>;
  using JacobianXjOplusType = Eigen::Matrix<double, D, VertexXj::Dimension>;
  static const int Dimension = D;

  BaseBinaryEdge() {
    _vertices.resize(2, nullptr);
    _information = InformationType::Identity();
  }
  int dimension() const override { return D; }
  const Measurement& measurement() const { return _measurement; }
  virtual void setMeasurement(const Measurement& m) { _measurement = m; }
  const InformationType& information() const { return _information; }
  InformationType& information() { return _information; }
  void setInformation(const InformationType& i) { _information = i; }
  const ErrorVector& error() const { return _error; }
  ErrorVector& error() { return _error; }
  double chi2() const override {
    double s = 0;
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) s += _error[r] * _information(r, c) * _error[c];
    return s;
  }
  const JacobianXiOplusType& jacobianOplusXi() const { return _jacobianOplusXi; }
  const JacobianXjOplusType& jacobianOplusXj() const { return _jacobianOplusXj; }

  // numeric Jacobian by central differences (g2o's default when a subclass only defines
  // computeError, e.g. src/sparse_gslam/src/g2o_bindings/edge_se2_rhotheta.cpp:9-16)
  void linearizeOplus() override {
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    ErrorVector errorBak = _error;
    for (int side = 0; side < 2; ++side) {
      OptimizableGraph::Vertex* v = static_cast<OptimizableGraph::Vertex*>(_vertices[side]);
      if (v->fixed()) continue;
      const int dim = v->dimension();
      std::vector<double> add(dim, 0.0);
      for (int d = 0; d < dim; ++d) {
        v->push();
        add[d] = delta;
        v->oplus(add.data());
        computeError();
        ErrorVector e1 = _error;
        v->pop();
        v->push();
        add[d] = -delta;
        v->oplus(add.data());
        computeError();
        ErrorVector e2 = _error;
        v->pop();
        add[d] = 0.0;
        for (int r = 0; r < D; ++r) {
          const double val = scalar * (e1[r] - e2[r]);
          if (side == 0) _jacobianOplusXi(r, d) = val;
          else _jacobianOplusXj(r, d) = val;
        }
      }
    }
    _error = errorBak;
  }
  // BaseBinaryEdge::constructQuadraticForm (+ robustInformation with the second-order term
  // disabled, as upstream): b -= J^T (rho1 Omega) e ; H += J^T (rho1 Omega) J
  void constructQuadraticForm(double* H, int n, double* b) override {
    OptimizableGraph::Vertex* from = static_cast<OptimizableGraph::Vertex*>(_vertices[0]);
    OptimizableGraph::Vertex* to = static_cast<OptimizableGraph::Vertex*>(_vertices[1]);
    const bool fromNotFixed = !from->fixed(), toNotFixed = !to->fixed();
    if (!fromNotFixed && !toNotFixed) return;
    double w = 1.0;
    if (robustKernel()) {
      Vector3 rho;
      robustKernel()->robustify(chi2(), rho);
      w = rho[1];
    }
    const int Di = VertexXi::Dimension, Dj = VertexXj::Dimension;
    double Oe[D];
    for (int r = 0; r < D; ++r) {
      double s = 0;
      for (int c = 0; c < D; ++c) s += _information(r, c) * _error[c];
      Oe[r] = w * s;
    }
    const int ci = from->colInHessian(), cj = to->colInHessian();
    auto JtOJ = [&](auto& Ja, int da, int ca, auto& Jb, int db, int cb) {
      for (int p = 0; p < da; ++p)
        for (int q = 0; q < db; ++q) {
          double s = 0;
          for (int r = 0; r < D; ++r)
            for (int c = 0; c < D; ++c) s += Ja(r, p) * w * _information(r, c) * Jb(c, q);
          H[(size_t)(ca + p) * n + cb + q] += s;
        }
    };
    if (fromNotFixed) {
      for (int p = 0; p < Di; ++p) {
        double s = 0;
        for (int r = 0; r < D; ++r) s += _jacobianOplusXi(r, p) * Oe[r];
        b[ci + p] -= s;
      }
      JtOJ(_jacobianOplusXi, Di, ci, _jacobianOplusXi, Di, ci);
    }
    if (toNotFixed) {
      for (int p = 0; p < Dj; ++p) {
        double s = 0;
        for (int r = 0; r < D; ++r) s += _jacobianOplusXj(r, p) * Oe[r];
        b[cj + p] -= s;
      }
      JtOJ(_jacobianOplusXj, Dj, cj, _jacobianOplusXj, Dj, cj);
    }
    if (fromNotFixed && toNotFixed) {
      JtOJ(_jacobianOplusXi, Di, ci, _jacobianOplusXj, Dj, cj);
      JtOJ(_jacobianOplusXj, Dj, cj, _jacobianOplusXi, Di, ci);
    }
  }
  virtual bool read(std::istream& is) = 0;
  virtual bool write(std::ostream& os) const = 0;

 protected:
  Measurement _measurement;
  InformationType _information;
  ErrorVector _error;
  JacobianXiOplusType _jacobianOplusXi;
  JacobianXjOplusType _jacobianOplusXj;
};

// ------------------------------------------------------------------------------- slam2d types
class VertexSE2 : public BaseVertex<3, SE2> {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  VertexSE2() {}
  void setToOriginImpl() override { _estimate = SE2(); }
  void oplusImpl(const double* update) override {
    Vector2 t = _estimate.translation();
    t[0] += update[0];
    t[1] += update[1];
    double angle = normalize_theta(_estimate.rotation().angle() + update[2]);
    _estimate.setTranslation(t);
    _estimate.setRotation(Rotation2D(angle));
  }
  bool read(std::istream& is) override {
    double x, y, t;
    is >> x >> y >> t;
    _estimate = SE2(x, y, t);
    return true;
  }
  bool write(std::ostream& os) const override {
    os << _estimate[0] << " " << _estimate[1] << " " << _estimate[2];
    return os.good();
  }
};

class EdgeSE2 : public BaseBinaryEdge<3, SE2, VertexSE2, VertexSE2> {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  EdgeSE2() {}
  void computeError() override {
    const VertexSE2* v1 = static_cast<const VertexSE2*>(_vertices[0]);
    const VertexSE2* v2 = static_cast<const VertexSE2*>(_vertices[1]);
    SE2 delta = _inverseMeasurement * (v1->estimate().inverse() * v2->estimate());
    _error = delta.toVector();
  }
  void setMeasurement(const SE2& m) override {
    _measurement = m;
    _inverseMeasurement = m.inverse();
  }
  const SE2& inverseMeasurement() const { return _inverseMeasurement; }
  void linearizeOplus() override {
    const VertexSE2* vi = static_cast<const VertexSE2*>(_vertices[0]);
    const VertexSE2* vj = static_cast<const VertexSE2*>(_vertices[1]);
    double thetai = vi->estimate().rotation().angle();
    Vector2 dt = vj->estimate().translation() - vi->estimate().translation();
    double si = std::sin(thetai), ci = std::cos(thetai);
    Matrix3 a, b, z;
    a << -ci, -si, -si * dt[0] + ci * dt[1], si, -ci, -ci * dt[0] - si * dt[1], 0, 0, -1;
    b << ci, si, 0, -si, ci, 0, 0, 0, 1;
    const double tz = _inverseMeasurement.rotation().angle();
    z << std::cos(tz), -std::sin(tz), 0, std::sin(tz), std::cos(tz), 0, 0, 0, 1;
    _jacobianOplusXi = z * a;
    _jacobianOplusXj = z * b;
  }
  // g2o text format of EDGE_SE2 (after the tag and the two vertex ids): dx dy dtheta, then the upper triangle
  // of the information matrix row by row (o11 o12 o13 o22 o23 o33)
  bool read(std::istream& is) override {
    double x, y, t;
    is >> x >> y >> t;
    setMeasurement(SE2(x, y, t));
    for (int i = 0; i < 3; ++i)
      for (int j = i; j < 3; ++j) {
        double v;
        is >> v;
        _information(i, j) = v;
        if (i != j) _information(j, i) = v;
      }
    return !is.fail();
  }
  bool write(std::ostream& os) const override {
    os << _measurement[0] << " " << _measurement[1] << " " << _measurement[2];
    for (int i = 0; i < 3; ++i)
      for (int j = i; j < 3; ++j) os << " " << _information(i, j);
    return os.good();
  }

 protected:
  SE2 _inverseMeasurement;
};

// ------------------------------------------------------------------------------- solver stack
// Only the configuration chain of graphs.cpp:9-23 is mirrored; the objects carry no state.
template <typename MatrixType>
class LinearSolver {
 public:
  virtual ~LinearSolver() {}
};
template <typename MatrixType>
class LinearSolverEigen : public LinearSolver<MatrixType> {};
class Solver {
 public:
  virtual ~Solver() {}
};
template <int P, int L>
struct BlockSolverTraits {
  static const int PoseDim = P;
  static const int LandmarkDim = L;
  struct PoseMatrixType {};
  using LinearSolverType = LinearSolver<PoseMatrixType>;
};
template <typename Traits>
class BlockSolver : public Solver {
 public:
  using PoseMatrixType = typename Traits::PoseMatrixType;
  using LinearSolverType = typename Traits::LinearSolverType;
  explicit BlockSolver(std::unique_ptr<LinearSolverType> ls) : _ls(std::move(ls)) {}

 private:
  std::unique_ptr<LinearSolverType> _ls;
};
class OptimizationAlgorithm {
 public:
  enum Kind { GaussNewton, Levenberg };
  explicit OptimizationAlgorithm(Kind k, std::unique_ptr<Solver> s) : _kind(k), _solver(std::move(s)) {}
  virtual ~OptimizationAlgorithm() {}
  Kind kind() const { return _kind; }

 private:
  Kind _kind;
  std::unique_ptr<Solver> _solver;
};
class OptimizationAlgorithmGaussNewton : public OptimizationAlgorithm {
 public:
  explicit OptimizationAlgorithmGaussNewton(std::unique_ptr<Solver> s) : OptimizationAlgorithm(GaussNewton, std::move(s)) {}
};
class OptimizationAlgorithmLevenberg : public OptimizationAlgorithm {
 public:
  explicit OptimizationAlgorithmLevenberg(std::unique_ptr<Solver> s) : OptimizationAlgorithm(Levenberg, std::move(s)) {}
};

// ------------------------------------------------------------------------------- optimiser
class SparseOptimizer : public OptimizableGraph {
 public:
  SparseOptimizer() {}
  ~SparseOptimizer() {
    if (_ctx) sgo_destroy(_ctx);  // never touches the algorithm, vertices or edges (caller-owned)
  }
  SparseOptimizer(const SparseOptimizer&) = delete;
  SparseOptimizer& operator=(const SparseOptimizer&) = delete;

  void setAlgorithm(OptimizationAlgorithm* a) { _algorithm = a; }
  OptimizationAlgorithm* algorithm() const { return _algorithm; }
  void setVerbose(bool v) { _verbose = v; }
  bool verbose() const { return _verbose; }
  void setComputeBatchStatistics(bool) {}

  // ---- container (OptimizableGraph / HyperGraph)
  bool addVertex(HyperGraph::Vertex* v) {
    if (!v || v->id() < 0) return false;
    if (_vertices.find(v->id()) != _vertices.end()) return false;
    _vertices[v->id()] = v;
    return true;
  }
  bool addEdge(HyperGraph::Edge* e) {
    if (!e) return false;
    for (auto* v : e->vertices())
      if (!v) return false;
    if (!_edges.insert(e).second) return false;
    e->_internalId = _nextEdgeId++;
    for (auto* v : e->vertices()) v->edges().insert(e);
    return true;
  }
  bool removeEdge(HyperGraph::Edge* e) {
    auto it = _edges.find(e);
    if (it == _edges.end()) return false;
    _edges.erase(it);
    for (auto* v : e->vertices())
      if (v) v->edges().erase(e);
    return true;
  }
  bool removeVertex(HyperGraph::Vertex* v, bool = false) {
    auto it = _vertices.find(v->id());
    if (it == _vertices.end() || it->second != v) return false;
    std::set<HyperGraph::Edge*> tmp(v->edges());
    for (auto* e : tmp) removeEdge(e);
    _vertices.erase(it);
    return true;
  }
  OptimizableGraph::Vertex* vertex(int id) {
    auto it = _vertices.find(id);
    return it == _vertices.end() ? nullptr : static_cast<OptimizableGraph::Vertex*>(it->second);
  }
  const VertexIDMap& vertices() const { return _vertices; }
  VertexIDMap& vertices() { return _vertices; }
  const EdgeSet& edges() const { return _edges; }
  EdgeSet& edges() { return _edges; }
  void clear() {  // upstream semantics: the vertices' own edge sets are left alone
    _vertices.clear();
    _edges.clear();
    _activeVertices.clear();
    _activeEdges.clear();
    _graphOnDevice = false;
  }

  // ---- optimisation
  bool initializeOptimization(int level = 0) {
    (void)level;
    _activeVertices.clear();
    _activeEdges.clear();
    // upstream rule: walk the edge sets OF THE VERTICES that are in the graph (not the graph's own
    // edge container) and activate every edge all of whose vertices are in the graph.  After
    // clear() a re-added vertex still lists its old edges (slc.cpp:259-261); they come back to
    // life only if every endpoint is back in the graph too.
    std::set<HyperGraph::Vertex*> vs;
    std::set<HyperGraph::Edge*> seen;
    std::vector<HyperGraph::Edge*> es;
    for (auto& kv : _vertices) {
      for (auto* e : kv.second->edges()) {
        if (!seen.insert(e).second) continue;
        bool all = true;
        for (auto* v : e->vertices()) {
          auto it = v ? _vertices.find(v->id()) : _vertices.end();
          if (it == _vertices.end() || it->second != v) all = false;
        }
        if (!all) continue;
        // upstream: `allVerticesOK && !e->allVerticesFixed()` -- an edge between fixed vertices is not
        // active (it would count in activeChi2 / activeEdges, drone.cpp:162-165, and nowhere else)
        bool allFixed = true;
        for (auto* v : e->vertices()) allFixed = allFixed && static_cast<OptimizableGraph::Vertex*>(v)->fixed();
        if (allFixed) continue;
        es.push_back(e);
        for (auto* v : e->vertices()) vs.insert(v);
      }
    }
    std::sort(es.begin(), es.end(), [](HyperGraph::Edge* a, HyperGraph::Edge* b) { return a->internalId() < b->internalId(); });
    for (auto* e : es) _activeEdges.push_back(static_cast<OptimizableGraph::Edge*>(e));
    for (auto* v : vs) _activeVertices.push_back(static_cast<OptimizableGraph::Vertex*>(v));
    std::sort(_activeVertices.begin(), _activeVertices.end(),
              [](OptimizableGraph::Vertex* a, OptimizableGraph::Vertex* b) { return a->id() < b->id(); });
    int h = 0;
    for (auto* v : _activeVertices) v->setHessianIndex(v->fixed() ? -1 : h++);
    // (what the device holds stays known: optimize() marshals the active sets again and compares -- an unchanged graph
    // uploads poses only, a graph that grew by a chain and a closure becomes an incremental update)
    return true;
  }
  // upstream appends the new vertices / edges to the active sets (new vertices get the next hessian indices)
  // and extends the solver's structure; vertices and edges already active keep their place
  bool updateInitialization(HyperGraph::VertexSet& vset, HyperGraph::EdgeSet& eset) {
    std::vector<OptimizableGraph::Vertex*> nv;
    for (auto* hv : vset) {
      auto* v = static_cast<OptimizableGraph::Vertex*>(hv);
      if (std::find(_activeVertices.begin(), _activeVertices.end(), v) == _activeVertices.end()) nv.push_back(v);
    }
    std::vector<OptimizableGraph::Edge*> ne;
    for (auto* he : eset) {
      auto* e = static_cast<OptimizableGraph::Edge*>(he);
      bool all = true, allFixed = true;
      for (auto* v : e->vertices()) {
        auto it = v ? _vertices.find(v->id()) : _vertices.end();
        if (it == _vertices.end() || it->second != v) all = false;
        else allFixed = allFixed && static_cast<OptimizableGraph::Vertex*>(v)->fixed();
      }
      if (!all || allFixed) continue;
      if (std::find(_activeEdges.begin(), _activeEdges.end(), e) == _activeEdges.end()) ne.push_back(e);
      for (auto* v : e->vertices()) {   // an edge activates its vertices (upstream asserts they are in vset)
        auto* ov = static_cast<OptimizableGraph::Vertex*>(v);
        if (std::find(_activeVertices.begin(), _activeVertices.end(), ov) == _activeVertices.end() &&
            std::find(nv.begin(), nv.end(), ov) == nv.end())
          nv.push_back(ov);
      }
    }
    std::sort(nv.begin(), nv.end(), [](OptimizableGraph::Vertex* a, OptimizableGraph::Vertex* b) { return a->id() < b->id(); });
    std::sort(ne.begin(), ne.end(), [](OptimizableGraph::Edge* a, OptimizableGraph::Edge* b) { return a->internalId() < b->internalId(); });
    int h = 0;
    for (auto* v : _activeVertices) h += v->fixed() ? 0 : 1;
    for (auto* v : nv) {
      v->setHessianIndex(v->fixed() ? -1 : h++);
      _activeVertices.push_back(v);
    }
    for (auto* e : ne) _activeEdges.push_back(e);
    return true;
  }

  // `online` (drone.cpp:155 passes true after updateInitialization): upstream it only decides whether the block
  // structure is rebuilt (BlockSolver::init / buildStructure at iteration 0) or was already extended by
  // updateInitialization -> updateStructure; the numbers are the same either way -- lambda is re-initialised from
  // the diagonal at iteration 0 of every optimize() call in both modes (OptimizationAlgorithmLevenberg::solve).
  // This backend marshals / assembles from the active sets on every call, so the flag has nothing left to select.
  int optimize(int iterations, bool online = false) {
    (void)online;
    if (!_algorithm) {
      std::cerr << "SparseOptimizer::optimize: no algorithm set" << std::endl;
      return -1;
    }
    bool anyFree = false;
    for (auto* v : _activeVertices) anyFree = anyFree || !v->fixed();
    if (_activeVertices.empty() || !anyFree) return -1;
    if (!gpuEligible()) return hostOptimize(iterations, online);
    if (!uploadGraph()) return 0;
    sgo_stats* st = new sgo_stats();
    int done = sgo_optimize_gn(_ctx, iterations, st);
    if (done < 0 && done != SGO_ENOTHING) {
      std::cerr << "SparseOptimizer::optimize: " << sgo_last_error(_ctx) << std::endl;
      done = 0;
    } else if (done == 0 && iterations > 0) {   // OptimizationAlgorithm::Fail: optimize() returns 0 upstream too
      std::cerr << "SparseOptimizer::optimize: linear solve failed after " << st->iters_done << " iterations: "
                << sgo_last_error(_ctx) << std::endl;
    }
    if (_verbose)
      for (int k = 0; k < st->iters_done; ++k)
        std::cerr << "iteration= " << k << "\t chi2= " << st->chi2[k + 1] << "\t time= " << st->seconds[k]
                  << "\t pcg= " << st->pcg_iters[k] << std::endl;
    _lastStats.reset(st);
    downloadEstimates();
    return done;
  }
  // ---- graph files (OptimizableGraph::load / save of g2o, for the types of this backend's device path):
  //   VERTEX_SE2 id x y theta | EDGE_SE2 i j dx dy dtheta o11 o12 o13 o22 o23 o33 | FIX id...
  // load() creates the objects and keeps them alive for the optimiser's lifetime (everything the CALLER adds
  // stays caller-owned, README.md:22-23); records of other types are skipped with a note on std::cerr, as
  // upstream does for unregistered tags.  save() writes the vertices in ascending id, FIX lines, then the edges
  // in insertion order, with max_digits10 precision (a load / save / load round trip is exact).
  bool load(std::istream& is) {
    std::string line;
    int skipped = 0;
    while (std::getline(is, line)) {
      std::istringstream ls(line);
      std::string tag;
      if (!(ls >> tag) || tag[0] == '#') continue;
      if (tag == "VERTEX_SE2") {
        int id;
        if (!(ls >> id)) return false;
        _loadedVertices.emplace_back();
        VertexSE2& v = _loadedVertices.back();
        v.setId(id);
        if (!v.read(ls) || ls.fail() || !addVertex(&v)) return false;
      } else if (tag == "EDGE_SE2") {
        int i, j;
        if (!(ls >> i >> j)) return false;
        auto* vi = vertex(i);
        auto* vj = vertex(j);
        if (!vi || !vj) {
          std::cerr << "SparseOptimizer::load: EDGE_SE2 " << i << " " << j << " references an unknown vertex" << std::endl;
          return false;
        }
        _loadedEdges.emplace_back();
        EdgeSE2& e = _loadedEdges.back();
        e.vertices()[0] = vi;
        e.vertices()[1] = vj;
        if (!e.read(ls) || !addEdge(&e)) return false;
      } else if (tag == "FIX") {
        int id;
        while (ls >> id)
          if (auto* v = vertex(id)) v->setFixed(true);
      } else {
        ++skipped;
      }
    }
    if (skipped) std::cerr << "SparseOptimizer::load: skipped " << skipped << " records of types this backend does not read" << std::endl;
    return true;
  }
  bool load(const char* filename) {
    std::ifstream f(filename);
    return f.good() && load(f);
  }
  bool save(std::ostream& os, int level = 0) const {
    (void)level;
    os << std::setprecision(17);
    std::vector<const OptimizableGraph::Vertex*> vs;
    for (auto& kv : _vertices) vs.push_back(static_cast<const OptimizableGraph::Vertex*>(kv.second));
    std::sort(vs.begin(), vs.end(), [](const OptimizableGraph::Vertex* a, const OptimizableGraph::Vertex* b) { return a->id() < b->id(); });
    for (auto* v : vs) {
      if (typeid(*v) != typeid(VertexSE2)) continue;
      os << "VERTEX_SE2 " << v->id() << " ";
      static_cast<const VertexSE2*>(v)->write(os);
      os << "\n";
    }
    for (auto* v : vs)
      if (v->fixed()) os << "FIX " << v->id() << "\n";
    std::vector<const HyperGraph::Edge*> es(_edges.begin(), _edges.end());
    std::sort(es.begin(), es.end(), [](const HyperGraph::Edge* a, const HyperGraph::Edge* b) { return a->internalId() < b->internalId(); });
    for (auto* he : es) {
      auto* e = static_cast<const OptimizableGraph::Edge*>(he);
      if (typeid(*e) != typeid(EdgeSE2)) continue;
      os << "EDGE_SE2 " << e->vertices()[0]->id() << " " << e->vertices()[1]->id() << " ";
      static_cast<const EdgeSE2*>(e)->write(os);
      os << "\n";
    }
    return os.good();
  }
  bool save(const char* filename, int level = 0) const {
    std::ofstream f(filename);
    return f.good() && save(f, level);
  }

  const sgo_stats* lastStats() const { return _lastStats.get(); }
  // which solver / set-up the backend used for the graph it holds (extension for the tests: sgo_solver_description)
  std::string backendDescription() const { return _ctx ? std::string(sgo_solver_description(_ctx)) : std::string(); }
  // Levenberg-Marquardt bookkeeping of the last optimize() on the host solver (extension for the tests; g2o
  // exposes the same numbers through OptimizationAlgorithmLevenberg::currentLambda() / levenbergIterations()
  // and its verbose output): damping after the iteration, robust chi2 after it, trials it took.
  struct LmIteration {
    double lambda, chi2;
    int trials;
  };
  const std::vector<LmIteration>& lmTrace() const { return _lmTrace; }

  void computeActiveErrors() {
    for (auto* e : _activeEdges) e->computeError();
  }
  double activeChi2() const {
    double s = 0;
    for (auto* e : _activeEdges) s += e->chi2();
    return s;
  }
  double activeRobustChi2() const {
    double s = 0;
    Vector3 rho;
    for (auto* e : _activeEdges) {
      if (e->robustKernel()) {
        e->robustKernel()->robustify(e->chi2(), rho);
        s += rho[0];
      } else {
        s += e->chi2();
      }
    }
    return s;
  }
  const std::vector<OptimizableGraph::Edge*>& activeEdges() const { return _activeEdges; }
  const std::vector<OptimizableGraph::Vertex*>& activeVertices() const { return _activeVertices; }

  void push() {
    for (auto* v : _activeVertices) v->push();
  }
  void pop() {
    for (auto* v : _activeVertices) v->pop();
  }
  void discardTop() {
    for (auto* v : _activeVertices) v->discardTop();
  }

 private:
  // ---- small dense host solver for graphs that are not the SE(2) GN pose graph -----------------
  // OptimizationAlgorithmGaussNewton::solve / OptimizationAlgorithmLevenberg::solve of g2o
  // 2020.5.29 on dense normal equations (the landmark graph has <= a few hundred unknowns).
  bool hostBuildSystem(std::vector<double>& H, std::vector<double>& b, int n) {
    std::fill(H.begin(), H.end(), 0.0);
    std::fill(b.begin(), b.end(), 0.0);
    for (auto* e : _activeEdges) {
      e->linearizeOplus();
      e->constructQuadraticForm(H.data(), n, b.data());
    }
    return true;
  }
  static bool hostCholeskySolve(std::vector<double> A, const std::vector<double>& b, int n, std::vector<double>& x) {
    for (int j = 0; j < n; ++j) {                     // in-place lower Cholesky
      double d = A[(size_t)j * n + j];
      for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      if (!(d > 0.0) || !std::isfinite(d)) return false;
      d = std::sqrt(d);
      A[(size_t)j * n + j] = d;
      for (int i = j + 1; i < n; ++i) {
        double s = A[(size_t)i * n + j];
        for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s / d;
      }
    }
    x = b;
    for (int i = 0; i < n; ++i) {
      double s = x[i];
      for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * x[k];
      x[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = x[i];
      for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k];
      x[i] = s / A[(size_t)i * n + i];
    }
    for (double v : x)
      if (!std::isfinite(v)) return false;
    return true;
  }
  void hostUpdate(const std::vector<double>& x) {
    for (auto* v : _activeVertices)
      if (!v->fixed()) v->oplus(x.data() + v->colInHessian());
  }
  int hostOptimize(int iterations, bool online) {
    (void)online;
    int n = 0;
    for (auto* v : _activeVertices) {
      v->setColInHessian(v->fixed() ? -1 : n);
      if (!v->fixed()) n += v->dimension();
    }
    if (n == 0) return -1;
    // This dense host solver exists for the landmark graph (<= a few hundred unknowns).  It is not a
    // fallback for pose graphs: a large graph that is not (VertexSE2, EdgeSE2, no kernel or DCS,
    // Gauss-Newton) is refused instead of being solved in O(n^3) on the host.
    if (n > kHostSolverMaxUnknowns) {
      std::cerr << "SparseOptimizer::optimize: " << n << " unknowns in a graph the device path does not cover "
                << "(only VertexSE2 / EdgeSE2 with no kernel or RobustKernelDCS under Gauss-Newton is); refusing"
                << std::endl;
      return 0;
    }
    const bool lm = _algorithm->kind() == OptimizationAlgorithm::Levenberg;
    _lmTrace.clear();
    std::vector<double> H((size_t)n * n), b(n), x(n);
    int cjIterations = 0;
    bool ok = true, failed = false;
    for (int it = 0; it < iterations && ok; ++it) {
      computeActiveErrors();
      double currentChi = activeRobustChi2();
      hostBuildSystem(H, b, n);
      if (!lm) {  // Gauss-Newton: one undamped step
        if (!hostCholeskySolve(H, b, n, x)) {
          std::cerr << "SparseOptimizer::optimize: Cholesky failure, solving failed" << std::endl;
          failed = true;
          break;
        }
        hostUpdate(x);
        ++cjIterations;
        continue;
      }
      if (it == 0) {  // computeLambdaInit: tau * max diagonal, tau = 1e-5
        double maxDiag = 0;
        for (int j = 0; j < n; ++j) maxDiag = std::max(std::fabs(H[(size_t)j * n + j]), maxDiag);
        _lmLambda = 1e-5 * maxDiag;
        _lmNi = 2.0;
      }
      double rho = 0;
      int qmax = 0;
      do {
        push();
        std::vector<double> Hl(H);
        for (int j = 0; j < n; ++j) Hl[(size_t)j * n + j] += _lmLambda;
        const bool ok2 = hostCholeskySolve(Hl, b, n, x);
        double tempChi = std::numeric_limits<double>::max();
        double scale = 1e-3;
        if (ok2) {
          hostUpdate(x);
          computeActiveErrors();
          tempChi = activeRobustChi2();
          for (int j = 0; j < n; ++j) scale += x[j] * (_lmLambda * x[j] + b[j]);
        }
        rho = (currentChi - tempChi) / scale;
        if (rho > 0 && std::isfinite(tempChi) && ok2) {  // accept
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          const double scaleFactor = std::max(1. / 3., alpha);
          _lmLambda *= scaleFactor;
          _lmNi = 2;
          currentChi = tempChi;
          discardTop();
        } else {  // reject: restore and increase the damping
          _lmLambda *= _lmNi;
          _lmNi *= 2;
          pop();
          if (!std::isfinite(_lmLambda)) break;
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      ++cjIterations;
      _lmTrace.push_back({_lmLambda, currentChi, qmax});
      if (qmax == 10 || rho == 0 || !std::isfinite(_lmLambda)) ok = false;  // Terminate
    }
    if (failed) return 0;
    return cjIterations;
  }
  double _lmLambda = 0.0, _lmNi = 2.0;
  std::vector<LmIteration> _lmTrace;
  static constexpr int kHostSolverMaxUnknowns = 3000;

  bool gpuEligible() const {
    if (_algorithm->kind() != OptimizationAlgorithm::GaussNewton) return false;
    for (auto* v : _activeVertices)
      if (typeid(*v) != typeid(VertexSE2)) return false;
    for (auto* e : _activeEdges) {
      if (typeid(*e) != typeid(EdgeSE2)) return false;
      if (e->robustKernel() && typeid(*e->robustKernel()) != typeid(RobustKernelDCS)) return false;
    }
    return true;
  }
  // marshal the pointer graph into the flat arrays of sgo_set_graph_se2 (compact vertex numbering
  // in ascending id) and upload.  g2o rebuilds the system from the edge objects on every optimize(),
  // so measurement / information / fixed / kernel-delta changes made between two optimize() calls
  // without an initializeOptimization() must be seen: the arrays are marshalled every time and compared
  // with what the device holds.  Three outcomes:
  //   * identical graph: only the poses are uploaded (sgo_set_poses);
  //   * the device's graph is a PREFIX of the new one -- what the reference's loop closer produces: the graph it
  //     optimised before + a chain of new poses + one closure (slc.cpp:205-226, :272-287) -- : sgo_update_graph_se2,
  //     which keeps the resident structures when the appended part allows it (include/sgo.h);
  //   * anything else: sgo_set_graph_se2.
  bool uploadGraph() {
    if (!_ctx) {
      _ctx = sgo_create(-1, nullptr);
      if (!_ctx) {
        std::cerr << "SparseOptimizer: " << sgo_last_error(nullptr) << std::endl;
        return false;
      }
    }
    const int V = (int)_activeVertices.size(), E = (int)_activeEdges.size();
    std::vector<double> poses(3 * (size_t)V);
    for (int k = 0; k < V; ++k) {
      VertexSE2* v = static_cast<VertexSE2*>(_activeVertices[k]);
      poses[3 * k] = v->estimate()[0];
      poses[3 * k + 1] = v->estimate()[1];
      poses[3 * k + 2] = v->estimate()[2];
      v->setTempIndex(k);   // compact numbering in ascending id (the array index sgo uses)
    }
    std::vector<uint8_t> fixed(V);
    std::vector<int32_t> vid(V);
    for (int k = 0; k < V; ++k) {
      fixed[k] = _activeVertices[k]->fixed() ? 1 : 0;
      vid[k] = _activeVertices[k]->id();
    }
    std::vector<int32_t> ei(E), ej(E);
    std::vector<double> meas(3 * (size_t)E), info(6 * (size_t)E), phi(E);
    for (int k = 0; k < E; ++k) {
      const EdgeSE2* e = static_cast<const EdgeSE2*>(_activeEdges[k]);
      ei[k] = static_cast<const OptimizableGraph::Vertex*>(e->vertex(0))->tempIndex();
      ej[k] = static_cast<const OptimizableGraph::Vertex*>(e->vertex(1))->tempIndex();
      for (int q = 0; q < 3; ++q) meas[3 * k + q] = e->measurement()[q];
      const auto& O = e->information();
      double* o = &info[6 * (size_t)k];
      o[0] = O(0, 0); o[1] = O(0, 1); o[2] = O(0, 2); o[3] = O(1, 1); o[4] = O(1, 2); o[5] = O(2, 2);
      phi[k] = e->robustKernel() ? e->robustKernel()->delta() : -1.0;
    }
    // is the device's graph (the previous marshal) a prefix of this one?  Same vertex ids and fixed flags for its
    // vertices (compact numbers then agree), bit-identical edge records for its edges
    auto same = [](const void* a, const void* b, size_t n) { return n == 0 || std::memcmp(a, b, n) == 0; };
    const int dV = (int)_m.vid.size(), dE = (int)_m.ei.size();
    const bool prefix = _graphOnDevice && V >= dV && E >= dE && same(vid.data(), _m.vid.data(), sizeof(int32_t) * (size_t)dV) &&
                        same(fixed.data(), _m.fixed.data(), (size_t)dV) && same(ei.data(), _m.ei.data(), sizeof(int32_t) * (size_t)dE) &&
                        same(ej.data(), _m.ej.data(), sizeof(int32_t) * (size_t)dE) && same(meas.data(), _m.meas.data(), sizeof(double) * 3 * (size_t)dE) &&
                        same(info.data(), _m.info.data(), sizeof(double) * 6 * (size_t)dE) && same(phi.data(), _m.phi.data(), sizeof(double) * (size_t)dE);
    if (prefix && V == dV && E == dE) {
      if (sgo_set_poses(_ctx, poses.data()) == SGO_OK) return true;
      std::cerr << "SparseOptimizer: " << sgo_last_error(_ctx) << std::endl;
      return false;
    }
    _graphOnDevice = false;
    const int rc = prefix ? sgo_update_graph_se2(_ctx, V, poses.data(), fixed.data(), E, ei.data(), ej.data(), meas.data(), info.data(),
                                                 phi.data(), dE)
                          : sgo_set_graph_se2(_ctx, V, poses.data(), fixed.data(), E, ei.data(), ej.data(), meas.data(), info.data(),
                                              phi.data());
    if (rc != SGO_OK) {
      std::cerr << "SparseOptimizer: " << sgo_last_error(_ctx) << std::endl;
      return false;
    }
    _graphOnDevice = true;
    _m.vid.swap(vid);
    _m.fixed.swap(fixed);
    _m.ei.swap(ei);
    _m.ej.swap(ej);
    _m.meas.swap(meas);
    _m.info.swap(info);
    _m.phi.swap(phi);
    return true;
  }
  void downloadEstimates() {
    const int V = (int)_activeVertices.size();
    std::vector<double> poses(3 * (size_t)V);
    if (sgo_get_poses(_ctx, poses.data()) != SGO_OK) {
      std::cerr << "SparseOptimizer: " << sgo_last_error(_ctx) << std::endl;
      return;
    }
    for (int k = 0; k < V; ++k) {
      VertexSE2* v = static_cast<VertexSE2*>(_activeVertices[k]);
      if (!v->fixed()) v->setEstimate(SE2(poses[3 * k], poses[3 * k + 1], poses[3 * k + 2]));
    }
  }

  OptimizationAlgorithm* _algorithm = nullptr;
  bool _verbose = false;
  VertexIDMap _vertices;
  EdgeSet _edges;
  long long _nextEdgeId = 0;
  std::vector<OptimizableGraph::Vertex*> _activeVertices;
  std::vector<OptimizableGraph::Edge*> _activeEdges;
  sgo_ctx* _ctx = nullptr;
  bool _graphOnDevice = false;
  struct Marshalled {   // what the device holds, as it was marshalled
    std::vector<int32_t> vid, ei, ej;
    std::vector<uint8_t> fixed;
    std::vector<double> meas, info, phi;
  } _m;
  std::unique_ptr<sgo_stats> _lastStats;
  std::deque<VertexSE2> _loadedVertices;   // objects created by load(): the optimiser's own
  std::deque<EdgeSE2> _loadedEdges;
};

// One line of the CARMEN-style trajectory file the reference writes for the external metricEvaluator
// (src/sparse_gslam/src/log_runner.cpp:18-23, :258-268; datasets/eval.sh:2-3):
//   FLASER 0 x y theta x y theta t myhost t
inline void write_carmen_result_line(std::ostream& os, const SE2& estimate, double time) {
  const double x = estimate[0], y = estimate[1], theta = estimate[2];
  os << "FLASER 0 " << x << ' ' << y << ' ' << theta << ' ' << x << ' ' << y << ' ' << theta << ' ' << time << " myhost " << time
     << '\n';
}

}  // namespace g2o
