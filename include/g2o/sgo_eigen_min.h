// sgo_eigen_min.h -- the handful of Eigen types the g2o call sites of sparse-gslam touch, for
// builds where real Eigen is not installed (this image).  With Eigen present the compat layer
// includes <Eigen/Core>/<Eigen/Geometry> instead and this file is not used.
//
// Members provided = the ones SURVEY.md section 8(b) lists as used on g2o objects at the
// reference's call sites: fixed Matrix<double,R,C> with operator(), operator[], =, .noalias(),
// .inverse() (2x2 / 3x3), .transpose(), * + -, .norm(), .dot(), .cast<T>(), setZero, Zero,
// Identity, comma initialiser, ostream <<; Rotation2D with .angle(), .inverse(), * Vector2,
// .toRotationMatrix(), .cast<T>(); aligned_allocator; EIGEN_MAKE_ALIGNED_OPERATOR_NEW.
#pragma once
#include <cmath>
#include <cstddef>
#include <memory>
#include <ostream>

#define SGO_EIGEN_MIN 1
#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif

namespace Eigen {

template <class T>
using aligned_allocator = std::allocator<T>;
// Eigen::Ref<T> as the reference's ls_extractor/utils.h uses it -- a parameter type that binds a matrix (or, for Ref<const T>,
// a temporary) without a copy: with fixed-size value types a plain reference does the same
template <class T>
using Ref = T&;

template <class S, int R, int C>
class Matrix {
 public:
  using Scalar = S;
  enum { RowsAtCompileTime = R, ColsAtCompileTime = C };
  S m[R * C > 0 ? R * C : 1];

  Matrix() {
    for (int i = 0; i < R * C; ++i) m[i] = S(0);
  }
  template <int RR = R, int CC = C, class = typename std::enable_if<RR * CC == 2>::type>
  Matrix(S a, S b) {
    m[0] = a;
    m[1] = b;
  }
  template <int RR = R, int CC = C, class = typename std::enable_if<RR * CC == 3>::type>
  Matrix(S a, S b, S c) {
    m[0] = a;
    m[1] = b;
    m[2] = c;
  }
  explicit Matrix(const S* p) {
    for (int i = 0; i < R * C; ++i) m[i] = p[i];
  }

  static Matrix Zero() { return Matrix(); }
  static Matrix Identity() {
    Matrix r;
    for (int i = 0; i < (R < C ? R : C); ++i) r(i, i) = S(1);
    return r;
  }
  void setZero() { *this = Matrix(); }
  void setIdentity() { *this = Identity(); }
  static constexpr int rows() { return R; }
  static constexpr int cols() { return C; }
  static constexpr int size() { return R * C; }

  S& operator()(int r, int c) { return m[r * C + c]; }   // row-major storage
  const S& operator()(int r, int c) const { return m[r * C + c]; }
  S& operator()(int i) { return m[i]; }
  const S& operator()(int i) const { return m[i]; }
  S& operator[](int i) { return m[i]; }
  const S& operator[](int i) const { return m[i]; }
  S& x() { return m[0]; }
  S& y() { return m[1]; }
  const S& x() const { return m[0]; }
  const S& y() const { return m[1]; }
  S* data() { return m; }
  const S* data() const { return m; }

  Matrix& noalias() { return *this; }

  // comma initialiser:  M << a, b, c, ...;  (row-major order, as Eigen)
  struct Comma {
    Matrix& M;
    int k;
    Comma& operator,(S v) {
      M.m[k++] = v;
      return *this;
    }
  };
  Comma operator<<(S v) {
    m[0] = v;
    return Comma{*this, 1};
  }

  Matrix<S, C, R> transpose() const {
    Matrix<S, C, R> t;
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) t(c, r) = (*this)(r, c);
    return t;
  }
  template <class T>
  Matrix<T, R, C> cast() const {
    Matrix<T, R, C> t;
    for (int i = 0; i < R * C; ++i) t.m[i] = (T)m[i];
    return t;
  }
  S squaredNorm() const {
    S s = 0;
    for (int i = 0; i < R * C; ++i) s += m[i] * m[i];
    return s;
  }
  S norm() const { return std::sqrt(squaredNorm()); }
  S dot(const Matrix& o) const {
    S s = 0;
    for (int i = 0; i < R * C; ++i) s += m[i] * o.m[i];
    return s;
  }
  Matrix inverse() const {
    static_assert(R == C && (R == 2 || R == 3 || R == 1), "inverse(): 1x1, 2x2 and 3x3 only");
    Matrix r;
    const Matrix& a = *this;
    if (R == 1) {
      r.m[0] = S(1) / m[0];
    } else if (R == 2) {
      S det = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0), id = S(1) / det;
      r(0, 0) = a(1, 1) * id;
      r(0, 1) = -a(0, 1) * id;
      r(1, 0) = -a(1, 0) * id;
      r(1, 1) = a(0, 0) * id;
    } else {
      S c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1), c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2),
        c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
      S det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02, id = S(1) / det;
      r(0, 0) = c00 * id;
      r(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id;
      r(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
      r(1, 0) = c01 * id;
      r(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id;
      r(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
      r(2, 0) = c02 * id;
      r(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id;
      r(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
    }
    return r;
  }
  Matrix& operator+=(const Matrix& o) {
    for (int i = 0; i < R * C; ++i) m[i] += o.m[i];
    return *this;
  }
  Matrix& operator-=(const Matrix& o) {
    for (int i = 0; i < R * C; ++i) m[i] -= o.m[i];
    return *this;
  }
  Matrix& operator*=(S s) {
    for (int i = 0; i < R * C; ++i) m[i] *= s;
    return *this;
  }
};

template <class S, int R, int C>
Matrix<S, R, C> operator+(Matrix<S, R, C> a, const Matrix<S, R, C>& b) {
  return a += b;
}
template <class S, int R, int C>
Matrix<S, R, C> operator-(Matrix<S, R, C> a, const Matrix<S, R, C>& b) {
  return a -= b;
}
template <class S, int R, int C>
Matrix<S, R, C> operator-(Matrix<S, R, C> a) {
  for (int i = 0; i < R * C; ++i) a.m[i] = -a.m[i];
  return a;
}
template <class S, int R, int C>
Matrix<S, R, C> operator*(Matrix<S, R, C> a, S s) {
  return a *= s;
}
template <class S, int R, int C>
Matrix<S, R, C> operator*(S s, Matrix<S, R, C> a) {
  return a *= s;
}
template <class S, int R, int K, int C>
Matrix<S, R, C> operator*(const Matrix<S, R, K>& a, const Matrix<S, K, C>& b) {
  Matrix<S, R, C> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      S s = 0;
      for (int k = 0; k < K; ++k) s += a(i, k) * b(k, j);
      r(i, j) = s;
    }
  return r;
}
template <class S, int R, int C>
std::ostream& operator<<(std::ostream& os, const Matrix<S, R, C>& a) {
  for (int r = 0; r < R; ++r) {
    for (int c = 0; c < C; ++c) os << (c ? " " : "") << a(r, c);
    if (r + 1 < R) os << "\n";
  }
  return os;
}

using Vector2d = Matrix<double, 2, 1>;
using Vector3d = Matrix<double, 3, 1>;
using Vector2f = Matrix<float, 2, 1>;
using Vector3f = Matrix<float, 3, 1>;
using Matrix2d = Matrix<double, 2, 2>;
using Matrix3d = Matrix<double, 3, 3>;
using Matrix2f = Matrix<float, 2, 2>;
using Matrix3f = Matrix<float, 3, 3>;

template <class S>
class Rotation2D {
 public:
  Rotation2D() : a_(0) {}
  explicit Rotation2D(S a) : a_(a) {}
  S& angle() { return a_; }
  const S& angle() const { return a_; }
  Rotation2D inverse() const { return Rotation2D(-a_); }
  Matrix<S, 2, 2> toRotationMatrix() const {
    Matrix<S, 2, 2> r;
    const S c = std::cos(a_), s = std::sin(a_);
    r(0, 0) = c;
    r(0, 1) = -s;
    r(1, 0) = s;
    r(1, 1) = c;
    return r;
  }
  Matrix<S, 2, 1> operator*(const Matrix<S, 2, 1>& v) const {
    const S c = std::cos(a_), s = std::sin(a_);
    return Matrix<S, 2, 1>(c * v[0] - s * v[1], s * v[0] + c * v[1]);
  }
  Rotation2D operator*(const Rotation2D& o) const { return Rotation2D(a_ + o.a_); }
  template <class T>
  Rotation2D<T> cast() const {
    return Rotation2D<T>((T)a_);
  }

 private:
  S a_;
};
using Rotation2Dd = Rotation2D<double>;
using Rotation2Df = Rotation2D<float>;

}  // namespace Eigen
