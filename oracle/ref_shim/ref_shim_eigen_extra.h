// ref_shim_eigen_extra.h — the rest of the Eigen surface that glomap/estimators/cost_function.h and global_positioning.cc name,
// for compiling the reference's GLOBAL POSITIONING PROBLEM BUILDER into oracle/_ref (see ref_shim/ceres/ceres.h).
// Two kinds of things live here: (1) what the BATA functors evaluate, for T = double — Map<Matrix<T,3,1>>, cast<T>(), R^T v —
// implemented in plain doubles; (2) compile-only stand-ins for code in cost_function.h that global positioning never calls
// (the Fetzer focal-length costs: JacobiSVD, 3x3 products, Vector4d) — they abort when executed.
#pragma once
#include <array>
#include <cstdlib>

#include "ref_shim_types.h"

namespace Eigen {
enum { ComputeFullU = 1, ComputeFullV = 2 };
template <> struct RefShimCast3<double> {  // (T = double: the recording AutoDiffCostFunction evaluates in doubles)
  using type = Vector3d;
  static Vector3d make(const Vector3d& v) { return v; }
};

template <typename T, int R, int C>
struct Matrix;

template <>
struct Matrix<double, 3, 1> : Vector3d {
  Matrix() = default;
  Matrix(const Vector3d& o) : Vector3d(o) {}
  struct CommaInit {  // grav_est << a, b, c;
    Matrix* m;
    int i;
    CommaInit& operator,(double x) { m->v[i++] = x; return *this; }
  };
  CommaInit operator<<(double x) { v[0] = x; return CommaInit{this, 1}; }
};

inline Matrix3d to_rotation_matrix(const Quaterniond& q) { return q.toRotationMatrix(); }

template <typename T, int N>
struct VectorN {
  T v[N] = {};
  void setZero() { for (auto& x : v) x = T(0); }
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  template <typename U> VectorN<U, N> cast() const { VectorN<U, N> r; for (int i = 0; i < N; ++i) r.v[i] = U(v[i]); return r; }
};
template <typename T, int N> using Vector = VectorN<T, N>;
using Vector4d = VectorN<double, 4>;

template <typename M>
struct JacobiSVD {  // only named by the Fetzer focal-length helpers, which global positioning never runs
  JacobiSVD(const M&, int) { std::abort(); }
  Vector3d singularValues() const { return Vector3d(); }
  M matrixU() const { return M(); }
  M matrixV() const { return M(); }
};

template <typename M>
struct Map;
template <>
struct Map<Matrix<double, 3, 1>> {
  double* p;
  explicit Map(double* q) : p(q) {}
  Map& operator=(const Vector3d& o) { p[0] = o(0); p[1] = o(1); p[2] = o(2); return *this; }
};
template <>
struct Map<const Matrix<double, 3, 1>> {
  const double* p;
  explicit Map(const double* q) : p(q) {}
  operator Vector3d() const { return Vector3d(p[0], p[1], p[2]); }
  Vector3d operator-(const Map& o) const { return Vector3d(p[0] - o.p[0], p[1] - o.p[1], p[2] - o.p[2]); }
};
inline Vector3d operator-(const Vector3d& a, const Map<const Matrix<double, 3, 1>>& b) { return a - static_cast<Vector3d>(b); }
inline Vector3d operator*(const Matrix3d& R, const Map<const Matrix<double, 3, 1>>& x) { return R * static_cast<Vector3d>(x); }
}  // namespace Eigen
