// ORACLE (test infrastructure only — never linked into or called by the product path).
//
// orc_common.hpp — helpers of the multithreaded C++ CPU restatement of GLOMAP's estimators
// (oracle/csrc/orc_{ra,gp,ba}.cc).  Everything here is plain C++17 + OpenMP; no third-party code.
//
// Determinism: every floating-point reduction runs in an order that does NOT depend on the number
// of threads — sums over the observations of a camera / a track run serially inside the owner's
// loop iteration, long owner lists are cut into fixed-size chunks whose partial sums are combined
// serially, and global dot products are reduced over fixed 4096-element chunks.  Two runs with
// different OMP thread counts therefore return bit-identical results; the `order` switch of the
// solvers (reverse summation order) is the knob for rounding-sensitivity experiments instead.
#pragma once

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace orc {

using i64 = long long;

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 ld3(const double* p) { return {p[0], p[1], p[2]}; }
inline void st3(double* p, V3 a) {
  p[0] = a.x;
  p[1] = a.y;
  p[2] = a.z;
}

constexpr i64 kChunk = 4096;

// Deterministic sum of f(i), i in [0, n): fixed chunks, chunk results added in index order.
template <class F>
double chunked_sum(i64 n, F f) {
  const i64 nc = (n + kChunk - 1) / kChunk;
  std::vector<double> part(nc, 0.0);
#pragma omp parallel for schedule(static)
  for (i64 c = 0; c < nc; ++c) {
    const i64 a = c * kChunk, b = std::min(n, a + kChunk);
    double s = 0.0;
    for (i64 i = a; i < b; ++i) s += f(i);
    part[c] = s;
  }
  double s = 0.0;
  for (i64 c = 0; c < nc; ++c) s += part[c];
  return s;
}

template <class F>
double chunked_max(i64 n, F f) {
  const i64 nc = (n + kChunk - 1) / kChunk;
  std::vector<double> part(nc, 0.0);
#pragma omp parallel for schedule(static)
  for (i64 c = 0; c < nc; ++c) {
    const i64 a = c * kChunk, b = std::min(n, a + kChunk);
    double s = 0.0;
    for (i64 i = a; i < b; ++i) s = std::max(s, f(i));
    part[c] = s;
  }
  double s = 0.0;
  for (i64 c = 0; c < nc; ++c) s = std::max(s, part[c]);
  return s;
}

inline double vdot(const std::vector<double>& a, const std::vector<double>& b) {
  const double *pa = a.data(), *pb = b.data();
  return chunked_sum((i64)a.size(), [=](i64 i) { return pa[i] * pb[i]; });
}

// Owner-major lists (CSR) cut into chunks of at most kChunk entries: the sums over a long list
// (one intrinsics block shared by every image, a hub camera) are formed per chunk in parallel and
// combined serially per owner — same result for any thread count.
struct OwnerLists {
  i64 num_owners = 0;
  std::vector<i64> off;       // [num_owners + 1]
  std::vector<i64> idx;       // [M] entry ids grouped by owner, ascending inside an owner
  std::vector<i64> chunk_owner, chunk_begin, chunk_end;  // chunks, grouped by owner
  std::vector<i64> owner_chunk0;                          // [num_owners + 1] first chunk of an owner

  void build(i64 owners, i64 m, const int32_t* owner_of) {
    num_owners = owners;
    off.assign(owners + 1, 0);
    for (i64 k = 0; k < m; ++k) off[owner_of[k] + 1]++;
    for (i64 o = 0; o < owners; ++o) off[o + 1] += off[o];
    idx.resize(m);
    std::vector<i64> cur(off.begin(), off.end() - 1);
    for (i64 k = 0; k < m; ++k) idx[cur[owner_of[k]]++] = k;
    owner_chunk0.assign(owners + 1, 0);
    for (i64 o = 0; o < owners; ++o) {
      owner_chunk0[o] = (i64)chunk_owner.size();
      for (i64 a = off[o]; a < off[o + 1]; a += kChunk) {
        chunk_owner.push_back(o);
        chunk_begin.push_back(a);
        chunk_end.push_back(std::min(off[o + 1], a + kChunk));
      }
    }
    owner_chunk0[owners] = (i64)chunk_owner.size();
  }

  // out[owner][W] = sum over the owner's entries of f(entry, acc[W]) (f adds into acc).
  template <int W, class F>
  void reduce(double* out, bool reverse, F f) const {
    const i64 nc = (i64)chunk_owner.size();
    std::vector<double> part((size_t)nc * W);
#pragma omp parallel for schedule(dynamic, 16)
    for (i64 c = 0; c < nc; ++c) {
      double acc[W];
      for (int j = 0; j < W; ++j) acc[j] = 0.0;
      if (!reverse)
        for (i64 a = chunk_begin[c]; a < chunk_end[c]; ++a) f(idx[a], acc);
      else
        for (i64 a = chunk_end[c] - 1; a >= chunk_begin[c]; --a) f(idx[a], acc);
      for (int j = 0; j < W; ++j) part[(size_t)c * W + j] = acc[j];
    }
#pragma omp parallel for schedule(static)
    for (i64 o = 0; o < num_owners; ++o) {
      double acc[W];
      for (int j = 0; j < W; ++j) acc[j] = 0.0;
      for (i64 c = owner_chunk0[o]; c < owner_chunk0[o + 1]; ++c)
        for (int j = 0; j < W; ++j) acc[j] += part[(size_t)c * W + j];
      for (int j = 0; j < W; ++j) out[(size_t)o * W + j] = acc[j];
    }
  }
};

// In-place inverse of a symmetric positive definite n x n matrix (row-major, n <= 24: the 6 + 16 joint block of orc_ba_wide.cc) by Cholesky.
// Returns false when a pivot is not positive.
inline bool spd_inverse(double* A, int n) {
  double L[24 * 24];
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / d;
    }
  }
  // invert L (lower), then A^-1 = L^-T L^-1
  double Li[24 * 24];
  for (int i = 0; i < n * n; ++i) Li[i] = 0.0;
  for (int j = 0; j < n; ++j) {
    Li[j * n + j] = 1.0 / L[j * n + j];
    for (int i = j + 1; i < n; ++i) {
      double v = 0.0;
      for (int k = j; k < i; ++k) v -= L[i * n + k] * Li[k * n + j];
      Li[i * n + j] = v / L[i * n + i];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double v = 0.0;
      for (int k = i; k < n; ++k) v += Li[k * n + i] * Li[k * n + j];
      A[i * n + j] = v;
      A[j * n + i] = v;
    }
  return true;
}

struct Huber {  // ceres::HuberLoss(a), optionally inside ceres::ScaledLoss(scale)
  double a, scale;
  inline void eval(double s, double& rho0, double& rho1) const {
    const double b = a * a;
    if (s > b) {
      const double r = std::sqrt(s);
      rho0 = scale * (2.0 * a * r - b);
      rho1 = scale * std::max(2.2250738585072014e-308, a / r);
    } else {
      rho0 = scale * s;
      rho1 = scale;
    }
  }
};

}  // namespace orc
