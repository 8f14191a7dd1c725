// ref_shim_linalg.h — the part of Eigen that glomap/estimators/global_rotation_averaging.cc, math/rigid3d.cc,
// estimators/rotation_initializer.cc and math/tree.cc name, so that `make -C oracle ref` can compile THOSE FILES from
// /root/reference, unmodified, into oracle/_ref/libref_glomap_ra.so (test infrastructure; tests/test_oracle_ref.py holds
// oracle/ra.py to it).  Eigen, CHOLMOD, COLMAP and Boost do not exist in this image, so everything in this directory is a
// stand-in written for this purpose, none of it reference code:
//   VectorXd / ArrayXd            dynamic vector of doubles with the segment / array / asDiagonal expressions the file uses
//   SparseMatrix<double>, Triplet row lists; products with diagonals, vectors and each other, evaluated eagerly
//   CholmodSupernodalLLT          envelope Cholesky after reverse Cuthill-McKee; counts its factorisations (= IRLS iterations)
//   AngleAxis<double>             Eigen/src/Geometry/AngleAxis.h restated: from a quaternion (2 atan2(|vec|, |w|), axis
//                                 flipped for w < 0), from a matrix (through the quaternion), toRotationMatrix (Rodrigues)
// What is pinned by compiling the reference against this is the reference's own logic ABOVE these types: unknown layout, rows
// and weights of the linear system, residuals, the update on the manifold, the IRLS weights, the two convergence tests, the
// start from the spanning tree and the conversion between image and rig rotations.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "ref_shim_eigen_extra.h"

#ifndef EIGEN_PI
#define EIGEN_PI 3.141592653589793238462643383279502884197169399375105820974944592307816406L  // Eigen/src/Core/util/Macros.h
#endif

namespace ref_shim {
struct RaCounters {
  long lad_solves = 0, lad_admm_iterations = 0, llt_factorizations = 0;
};
inline RaCounters& ra_counters() {
  static RaCounters c;
  return c;
}
}  // namespace ref_shim

namespace Eigen {
struct ConstSeg {
  const double* p;
  long n;
  operator Vector3d() const { return Vector3d(p[0], p[1], p[2]); }
  Vector3d operator-() const { return Vector3d(-p[0], -p[1], -p[2]); }
  double squaredNorm() const { double s = 0; for (long i = 0; i < n; ++i) s += p[i] * p[i]; return s; }
  double norm() const { return std::sqrt(squaredNorm()); }
};
struct Seg : ConstSeg {
  double* q;
  Seg(double* d, long len) : ConstSeg{d, len}, q(d) {}
  Seg& operator=(const Vector3d& v) { q[0] = v(0); q[1] = v(1); q[2] = v(2); return *this; }
  void setConstant(double c) { for (long i = 0; i < n; ++i) q[i] = c; }
};
struct DiagonalWrapper {
  const std::vector<double>* d;
};
struct ArrayOps {  // what `.array()` is asked for: isNaN().any(), abs().sum()
  std::vector<double> a;
  struct Flags { bool any_; bool any() const { return any_; } };
  Flags isNaN() const { bool f = false; for (double x : a) f = f || std::isnan(x); return Flags{f}; }
  ArrayOps abs() const { ArrayOps r; for (double x : a) r.a.push_back(std::fabs(x)); return r; }
  double sum() const { double s = 0; for (double x : a) s += x; return s; }
};
struct VectorXd {  // also stands for ArrayXd: `.matrix()` is the identity here
  std::vector<double> a;
  VectorXd() = default;
  explicit VectorXd(size_t n) : a(n, 0.0) {}
  static VectorXd Ones(size_t n) { VectorXd v(n); for (double& x : v.a) x = 1.0; return v; }
  static VectorXd Zero(size_t n) { return VectorXd(n); }
  void resize(size_t n) { a.assign(n, 0.0); }
  void conservativeResize(size_t n) { a.resize(n, 0.0); }
  long size() const { return static_cast<long>(a.size()); }
  long rows() const { return size(); }
  double& operator[](long i) { return a[static_cast<size_t>(i)]; }
  const double& operator[](long i) const { return a[static_cast<size_t>(i)]; }
  double& operator()(long i) { return a[static_cast<size_t>(i)]; }
  const double& operator()(long i) const { return a[static_cast<size_t>(i)]; }
  void setZero() { for (double& x : a) x = 0.0; }
  double squaredNorm() const { double s = 0; for (double x : a) s += x * x; return s; }
  double norm() const { return std::sqrt(squaredNorm()); }
  Seg segment(long i, long n) { return Seg(a.data() + i, n); }
  ConstSeg segment(long i, long n) const { return ConstSeg{a.data() + i, n}; }
  template <int N> Seg segment(long i) { return Seg(a.data() + i, N); }
  template <int N> ConstSeg segment(long i) const { return ConstSeg{a.data() + i, N}; }
  VectorXd operator-(const VectorXd& o) const { VectorXd r(a.size()); for (size_t i = 0; i < a.size(); ++i) r.a[i] = a[i] - o.a[i]; return r; }
  VectorXd operator+(const VectorXd& o) const { VectorXd r(a.size()); for (size_t i = 0; i < a.size(); ++i) r.a[i] = a[i] + o.a[i]; return r; }
  VectorXd operator*(double s) const { VectorXd r(a.size()); for (size_t i = 0; i < a.size(); ++i) r.a[i] = a[i] * s; return r; }
  VectorXd& operator+=(const VectorXd& o) { for (size_t i = 0; i < a.size(); ++i) a[i] += o.a[i]; return *this; }
  ArrayOps array() const { return ArrayOps{a}; }
  const VectorXd& matrix() const { return *this; }
  DiagonalWrapper asDiagonal() const { return DiagonalWrapper{&a}; }
};
inline VectorXd operator*(double s, const VectorXd& v) { return v * s; }
using ArrayXd = VectorXd;
inline VectorXd operator*(const DiagonalWrapper& D, const VectorXd& v) {
  VectorXd r(v.a.size());
  for (size_t i = 0; i < v.a.size(); ++i) r.a[i] = (*D.d)[i] * v.a[i];
  return r;
}

template <typename T>
struct Triplet {
  long r, c;
  T v;
  Triplet(long row, long col, T val) : r(row), c(col), v(val) {}
};

template <typename T>
struct SparseMatrix {  // row lists, columns ascending inside a row
  long nr = 0, nc = 0;
  std::vector<std::vector<std::pair<long, T>>> row;
  void resize(long r, long c) { nr = r; nc = c; row.assign(static_cast<size_t>(r), {}); }
  long rows() const { return nr; }
  long cols() const { return nc; }
  template <typename It>
  void setFromTriplets(It b, It e) {  // duplicates are summed (Eigen/src/SparseCore/SparseMatrix.h)
    for (auto& r : row) r.clear();
    for (It t = b; t != e; ++t) {
      auto& r = row[static_cast<size_t>(t->r)];
      bool found = false;
      for (auto& cv : r)
        if (cv.first == t->c) { cv.second += t->v; found = true; break; }
      if (!found) r.emplace_back(t->c, t->v);
    }
    for (auto& r : row) std::sort(r.begin(), r.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
  }
  SparseMatrix transpose() const {
    SparseMatrix t;
    t.resize(nc, nr);
    for (long i = 0; i < nr; ++i)
      for (const auto& cv : row[static_cast<size_t>(i)]) t.row[static_cast<size_t>(cv.first)].emplace_back(i, cv.second);
    return t;
  }
  VectorXd operator*(const VectorXd& x) const {
    VectorXd y(static_cast<size_t>(nr));
    for (long i = 0; i < nr; ++i) {
      T s = 0;
      for (const auto& cv : row[static_cast<size_t>(i)]) s += cv.second * x.a[static_cast<size_t>(cv.first)];
      y.a[static_cast<size_t>(i)] = s;
    }
    return y;
  }
  SparseMatrix operator*(const DiagonalWrapper& D) const {  // column scaling
    SparseMatrix r = *this;
    for (auto& rw : r.row)
      for (auto& cv : rw) cv.second *= (*D.d)[static_cast<size_t>(cv.first)];
    return r;
  }
  SparseMatrix operator*(const SparseMatrix& B) const {
    SparseMatrix C;
    C.resize(nr, B.nc);
    std::vector<T> acc(static_cast<size_t>(B.nc), T(0));
    std::vector<char> hit(static_cast<size_t>(B.nc), 0);
    std::vector<long> cols;
    for (long i = 0; i < nr; ++i) {
      cols.clear();
      for (const auto& ik : row[static_cast<size_t>(i)])
        for (const auto& kj : B.row[static_cast<size_t>(ik.first)]) {
          if (!hit[static_cast<size_t>(kj.first)]) { hit[static_cast<size_t>(kj.first)] = 1; cols.push_back(kj.first); }
          acc[static_cast<size_t>(kj.first)] += ik.second * kj.second;
        }
      std::sort(cols.begin(), cols.end());
      for (long j : cols) {
        C.row[static_cast<size_t>(i)].emplace_back(j, acc[static_cast<size_t>(j)]);
        acc[static_cast<size_t>(j)] = 0;
        hit[static_cast<size_t>(j)] = 0;
      }
    }
    return C;
  }
};
template <typename T>
inline SparseMatrix<T> operator*(const DiagonalWrapper& D, const SparseMatrix<T>& A) {  // row scaling
  SparseMatrix<T> r = A;
  for (long i = 0; i < r.nr; ++i)
    for (auto& cv : r.row[static_cast<size_t>(i)]) cv.second *= (*D.d)[static_cast<size_t>(i)];
  return r;
}

enum ComputationInfo { Success = 0, NumericalIssue = 1 };

// Cholesky of a symmetric positive definite matrix given as a SparseMatrix.  Round 6: an ENVELOPE (skyline) factorisation
// after a reverse Cuthill-McKee ordering of the pattern instead of a dense one, so that the reference's rotation averaging also
// runs at BASELINE configs[3] size here (30 000 unknowns; a ring view graph orders to a band of a few hundred) — CHOLMOD's
// supernodal factorisation computes the same factor up to rounding.  analyzePattern is a no-op (the ordering is computed with the
// first factorisation and kept while the pattern's size does not change); factorize counts itself.
template <typename M>
struct CholmodSupernodalLLT {
  long n = 0;
  std::vector<long> perm, inv;    // perm[new] = old
  std::vector<long> first, off;   // envelope: row i (new numbering) holds columns first[i] .. i at L[off[i] ...]
  std::vector<double> L;
  ComputationInfo info_ = Success;
  void analyzePattern(const M&) {}
  void compute(const M& A) { factorize(A); }
  void order(const M& A) {  // reverse Cuthill-McKee on the pattern, components in index order, neighbours by degree
    n = A.rows();
    std::vector<std::vector<long>> adj(static_cast<size_t>(n));
    for (long i = 0; i < n; ++i)
      for (const auto& cv : A.row[static_cast<size_t>(i)])
        if (cv.first != i) {
          adj[static_cast<size_t>(i)].push_back(cv.first);
          adj[static_cast<size_t>(cv.first)].push_back(i);
        }
    for (auto& a : adj) {
      std::sort(a.begin(), a.end());
      a.erase(std::unique(a.begin(), a.end()), a.end());
    }
    std::vector<char> seen(static_cast<size_t>(n), 0);
    std::vector<long> ord;
    ord.reserve(static_cast<size_t>(n));
    for (long s0 = 0; s0 < n; ++s0) {
      if (seen[static_cast<size_t>(s0)]) continue;
      // start from a node far from s0 (one BFS sweep to a last-level node of small degree)
      long start = s0;
      {
        std::vector<long> q{s0};
        std::vector<char> vis(static_cast<size_t>(n), 0);
        vis[static_cast<size_t>(s0)] = 1;
        for (size_t h = 0; h < q.size(); ++h)
          for (long w : adj[static_cast<size_t>(q[h])])
            if (!vis[static_cast<size_t>(w)] && !seen[static_cast<size_t>(w)]) { vis[static_cast<size_t>(w)] = 1; q.push_back(w); }
        start = q.back();
      }
      size_t h = ord.size();
      ord.push_back(start);
      seen[static_cast<size_t>(start)] = 1;
      for (; h < ord.size(); ++h) {
        std::vector<long> nb;
        for (long w : adj[static_cast<size_t>(ord[h])])
          if (!seen[static_cast<size_t>(w)]) { seen[static_cast<size_t>(w)] = 1; nb.push_back(w); }
        std::stable_sort(nb.begin(), nb.end(), [&](long x, long y) { return adj[static_cast<size_t>(x)].size() < adj[static_cast<size_t>(y)].size(); });
        for (long w : nb) ord.push_back(w);
      }
    }
    std::reverse(ord.begin(), ord.end());
    perm = ord;
    inv.assign(static_cast<size_t>(n), 0);
    for (long k = 0; k < n; ++k) inv[static_cast<size_t>(perm[static_cast<size_t>(k)])] = k;
  }
  void factorize(const M& A) {
    ++ref_shim::ra_counters().llt_factorizations;
    if (A.rows() != n || perm.empty()) order(A);
    first.assign(static_cast<size_t>(n), 0);
    for (long k = 0; k < n; ++k) {
      long f = k;
      for (const auto& cv : A.row[static_cast<size_t>(perm[static_cast<size_t>(k)])]) f = std::min(f, inv[static_cast<size_t>(cv.first)]);
      first[static_cast<size_t>(k)] = f;
    }
    // (a symmetric pattern is assumed for the envelope; entries above it would be the transposes of entries inside it)
    off.assign(static_cast<size_t>(n) + 1, 0);
    for (long k = 0; k < n; ++k) off[static_cast<size_t>(k) + 1] = off[static_cast<size_t>(k)] + (k - first[static_cast<size_t>(k)] + 1);
    if (std::getenv("REF_SHIM_LLT_DEBUG")) std::fprintf(stderr, "[llt] n %ld envelope %ld\n", n, off[static_cast<size_t>(n)]);
    L.assign(static_cast<size_t>(off[static_cast<size_t>(n)]), 0.0);
    for (long k = 0; k < n; ++k)
      for (const auto& cv : A.row[static_cast<size_t>(perm[static_cast<size_t>(k)])]) {
        const long j = inv[static_cast<size_t>(cv.first)];
        if (j <= k) L[static_cast<size_t>(off[static_cast<size_t>(k)] + j - first[static_cast<size_t>(k)])] = cv.second;
      }
    info_ = Success;
    for (long i = 0; i < n; ++i) {
      const long fi = first[static_cast<size_t>(i)];
      double* li = &L[static_cast<size_t>(off[static_cast<size_t>(i)])];
      for (long j = fi; j <= i; ++j) {
        const long fj = first[static_cast<size_t>(j)];
        const double* lj = &L[static_cast<size_t>(off[static_cast<size_t>(j)])];
        double s = li[j - fi];
        for (long k = std::max(fi, fj); k < j; ++k) s -= li[k - fi] * lj[k - fj];
        if (j < i) {
          li[j - fi] = s / lj[j - fj];
        } else {
          if (!(s > 0.0)) { info_ = NumericalIssue; s = std::nan(""); }
          li[j - fi] = std::sqrt(s);
        }
      }
    }
  }
  ComputationInfo info() const { return info_; }
  VectorXd solve(const VectorXd& b) const {
    std::vector<double> y(static_cast<size_t>(n));
    for (long k = 0; k < n; ++k) y[static_cast<size_t>(k)] = b.a[static_cast<size_t>(perm[static_cast<size_t>(k)])];
    for (long i = 0; i < n; ++i) {  // L y = b
      const long fi = first[static_cast<size_t>(i)];
      const double* li = &L[static_cast<size_t>(off[static_cast<size_t>(i)])];
      double s = y[static_cast<size_t>(i)];
      for (long k = fi; k < i; ++k) s -= li[k - fi] * y[static_cast<size_t>(k)];
      y[static_cast<size_t>(i)] = s / li[i - fi];
    }
    for (long i = n - 1; i >= 0; --i) {  // L^T x = y, column sweep over the row storage
      const long fi = first[static_cast<size_t>(i)];
      const double* li = &L[static_cast<size_t>(off[static_cast<size_t>(i)])];
      const double xi = y[static_cast<size_t>(i)] / li[i - fi];
      y[static_cast<size_t>(i)] = xi;
      for (long k = fi; k < i; ++k) y[static_cast<size_t>(k)] -= li[k - fi] * xi;
    }
    VectorXd x(static_cast<size_t>(n));
    for (long k = 0; k < n; ++k) x.a[static_cast<size_t>(perm[static_cast<size_t>(k)])] = y[static_cast<size_t>(k)];
    return x;
  }
};
template <typename M>
using SimplicialLLT = CholmodSupernodalLLT<M>;

// Eigen/src/Geometry/AngleAxis.h
template <typename T>
struct AngleAxis {
  double angle_ = 0.0;
  Vector3d axis_ = Vector3d(1, 0, 0);
  AngleAxis(double angle, const Vector3d& axis) : angle_(angle), axis_(axis) {}
  explicit AngleAxis(const Quaterniond& q) { from_quaternion(q); }
  explicit AngleAxis(const Matrix3d& m) { from_quaternion(Quaterniond(m)); }  // `return *this = QuaternionType(mat);`
  double angle() const { return angle_; }
  const Vector3d& axis() const { return axis_; }
  void from_quaternion(const Quaterniond& q) {  // operator=(const QuaternionBase&)
    double n = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
    // (n < epsilon: Eigen re-evaluates with stableNorm(); same value up to rounding)
    if (n != 0.0) {
      angle_ = 2.0 * std::atan2(n, std::fabs(q.w()));
      if (q.w() < 0.0) n = -n;
      axis_ = Vector3d(q.x() / n, q.y() / n, q.z() / n);
    } else {
      angle_ = 0.0;
      axis_ = Vector3d(1, 0, 0);
    }
  }
  Matrix3d toRotationMatrix() const {
    Matrix3d res;
    const double s = std::sin(angle_), c = std::cos(angle_);
    const Vector3d sin_axis = s * axis_;
    const Vector3d cos1_axis = (1.0 - c) * axis_;
    double tmp;
    tmp = cos1_axis(0) * axis_(1);
    res(0, 1) = tmp - sin_axis(2);
    res(1, 0) = tmp + sin_axis(2);
    tmp = cos1_axis(0) * axis_(2);
    res(0, 2) = tmp + sin_axis(1);
    res(2, 0) = tmp - sin_axis(1);
    tmp = cos1_axis(1) * axis_(2);
    res(1, 2) = tmp - sin_axis(0);
    res(2, 1) = tmp + sin_axis(0);
    res(0, 0) = cos1_axis(0) * axis_(0) + c;
    res(1, 1) = cos1_axis(1) * axis_(1) + c;
    res(2, 2) = cos1_axis(2) * axis_(2) + c;
    return res;
  }
};
}  // namespace Eigen
