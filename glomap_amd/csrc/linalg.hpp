// linalg.hpp — small fixed-size linear algebra and loss helpers shared by gp.hip and ba.hip.
#pragma once

#include <hip/hip_runtime.h>

namespace gsfm {

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 ld3(const double* __restrict__ p) { return V3{p[0], p[1], p[2]}; }
// Six doubles at a 16-byte aligned address as three 16-byte loads (one gather instruction each).
__device__ __forceinline__ void ld6(const double* __restrict__ p, V3& a, V3& b) {
  const double2* q = reinterpret_cast<const double2*>(p);
  const double2 v0 = q[0], v1 = q[1], v2 = q[2];
  a = V3{v0.x, v0.y, v1.x};
  b = V3{v1.y, v2.x, v2.y};
}
// First three doubles of a 16-byte aligned record (two 16-byte loads).
__device__ __forceinline__ V3 ld3a(const double* __restrict__ p) {
  const double2* q = reinterpret_cast<const double2*>(p);
  const double2 v0 = q[0], v1 = q[1];
  return V3{v0.x, v0.y, v1.x};
}
__device__ __forceinline__ void st3(double* __restrict__ p, const V3& v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}
__device__ __forceinline__ V3 operator+(const V3& a, const V3& b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, const V3& a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// symmetric 3x3: (xx, xy, xz, yy, yz, zz)
struct S3 {
  double xx, xy, xz, yy, yz, zz;
};
__device__ __forceinline__ V3 mul(const S3& m, const V3& v) {
  return V3{m.xx * v.x + m.xy * v.y + m.xz * v.z, m.xy * v.x + m.yy * v.y + m.yz * v.z,
            m.xz * v.x + m.yz * v.y + m.zz * v.z};
}
__device__ __forceinline__ S3 inv3(const S3& m) {
  const double c00 = m.yy * m.zz - m.yz * m.yz;
  const double c01 = m.xz * m.yz - m.xy * m.zz;
  const double c02 = m.xy * m.yz - m.xz * m.yy;
  const double det = m.xx * c00 + m.xy * c01 + m.xz * c02;
  const double id = 1.0 / det;
  return S3{c00 * id, c01 * id, c02 * id, (m.xx * m.zz - m.xz * m.xz) * id, (m.xy * m.xz - m.xx * m.yz) * id,
            (m.xx * m.yy - m.xy * m.xy) * id};
}
// Q v = a (v - beta d (d.v))
__device__ __forceinline__ V3 applyQ(double a, double beta, const V3& d, const V3& v) {
  const double k = beta * dot(d, v);
  return V3{a * (v.x - k * d.x), a * (v.y - k * d.y), a * (v.z - k * d.z)};
}

__device__ __forceinline__ void huber(double a, double scale, double sq, double& rho, double& w) {
  if (sq > a * a) {
    const double r = sqrt(sq);
    rho = scale * (2.0 * a * r - a * a);
    w = scale * (a / r);
  } else {
    rho = scale * sq;
    w = scale;
  }
}

__device__ __forceinline__ void atomic_add3(double* p, const V3& v) {
  unsafeAtomicAdd(p, v.x);
  unsafeAtomicAdd(p + 1, v.y);
  unsafeAtomicAdd(p + 2, v.z);
}


// In-place inverse of a symmetric positive definite n x n matrix (row-major, n <= 8) by Cholesky:
// A = L L^T, A^-1 = L^-T L^-1.  Returns false when a pivot is not positive.
template <int MAXN>
__device__ inline bool spd_inverse(double* A, int n) {
  double L[MAXN * MAXN];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j <= i; ++j) {
      double sum = A[i * n + j];
      for (int k = 0; k < j; ++k) sum -= L[i * MAXN + k] * L[j * MAXN + k];
      if (i == j) {
        if (!(sum > 0.0)) return false;
        L[i * MAXN + i] = sqrt(sum);
      } else {
        L[i * MAXN + j] = sum / L[j * MAXN + j];
      }
    }
  }
  // Linv (lower)
  double Li[MAXN * MAXN];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) Li[i * MAXN + j] = 0.0;
    Li[i * MAXN + i] = 1.0 / L[i * MAXN + i];
    for (int j = 0; j < i; ++j) {
      double sum = 0.0;
      for (int k = j; k < i; ++k) sum -= L[i * MAXN + k] * Li[k * MAXN + j];
      Li[i * MAXN + j] = sum / L[i * MAXN + i];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double sum = 0.0;
      for (int k = (i > j ? i : j); k < n; ++k) sum += Li[k * MAXN + i] * Li[k * MAXN + j];
      A[i * n + j] = sum;
    }
  return true;
}

}  // namespace gsfm
