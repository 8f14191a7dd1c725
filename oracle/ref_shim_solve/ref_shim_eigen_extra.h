// ref_shim_solve/ref_shim_eigen_extra.h — shadows ref_shim/ref_shim_eigen_extra.h (this directory comes first on the include path
// of the SOLVING reference libraries, oracle/Makefile `ref_solve`): the same Eigen surface of glomap/estimators/cost_function.h
// and global_positioning.cc, but for ANY scalar T, so that the reference's own BATA functors can be evaluated on the dual
// numbers (ceres::Jet) of the solving Ceres stand-in next door (ceres/ceres.h) — their Jacobians are then the derivatives of
// the reference's code, not of a restatement.  T = double keeps the plain-double types of ref_shim_types.h.
// Test infrastructure; nothing here is reference code.
#pragma once
#include <array>
#include <cstdlib>

#include "ref_shim_types.h"

namespace Eigen {
enum { ComputeFullU = 1, ComputeFullV = 2 };

template <typename T, int R, int C>
struct Matrix;

template <typename T>
struct Matrix<T, 3, 1> {  // three scalars of type T, evaluated left to right as written
  T v[3] = {T(0.0), T(0.0), T(0.0)};
  Matrix() = default;
  Matrix(const T& a, const T& b, const T& c) : v{a, b, c} {}
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  Matrix operator-(const Matrix& o) const { return Matrix(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Matrix operator+(const Matrix& o) const { return Matrix(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  T& operator[](int i) { return v[i]; }
  const T& operator[](int i) const { return v[i]; }
  struct CommaInit {  // grav_est << a, b, c;   (GravError: compiled, never run by global positioning)
    Matrix* m;
    int i;
    CommaInit& operator,(const T& x) { m->v[i++] = x; return *this; }
  };
  CommaInit operator<<(const T& x) { v[0] = x; return CommaInit{this, 1}; }
};
template <typename T>
Matrix<T, 3, 1> operator*(const T& s, const Matrix<T, 3, 1>& a) { return Matrix<T, 3, 1>(a.v[0] * s, a.v[1] * s, a.v[2] * s); }

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

template <typename T> struct RefShimCast3 {
  using type = Matrix<T, 3, 1>;
  static type make(const Vector3d& a) { return type(T(a(0)), T(a(1)), T(a(2))); }
};
template <> struct RefShimCast3<double> {
  using type = Vector3d;
  static Vector3d make(const Vector3d& a) { return a; }
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
template <typename T>
struct Map<Matrix<T, 3, 1>> {
  T* p;
  explicit Map(T* q) : p(q) {}
  Map& operator=(const Matrix<T, 3, 1>& o) { p[0] = o(0); p[1] = o(1); p[2] = o(2); return *this; }
};
template <typename T>
struct Map<const Matrix<T, 3, 1>> {
  const T* p;
  explicit Map(const T* q) : p(q) {}
  operator Matrix<T, 3, 1>() const { return Matrix<T, 3, 1>(p[0], p[1], p[2]); }
  Matrix<T, 3, 1> operator-(const Map& o) const { return Matrix<T, 3, 1>(p[0] - o.p[0], p[1] - o.p[1], p[2] - o.p[2]); }
};
template <typename T>
Matrix<T, 3, 1> operator-(const Matrix<T, 3, 1>& a, const Map<const Matrix<T, 3, 1>>& b) { return a - static_cast<Matrix<T, 3, 1>>(b); }
template <typename T>
Matrix<T, 3, 1> operator*(const Matrix3d& R, const Map<const Matrix<T, 3, 1>>& x) {  // a double matrix times a vector of T
  return Matrix<T, 3, 1>(R(0, 0) * x.p[0] + R(0, 1) * x.p[1] + R(0, 2) * x.p[2], R(1, 0) * x.p[0] + R(1, 1) * x.p[1] + R(1, 2) * x.p[2],
                         R(2, 0) * x.p[0] + R(2, 1) * x.p[1] + R(2, 2) * x.p[2]);
}
// T = double: the plain-double types, with the operation order of the original header
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
