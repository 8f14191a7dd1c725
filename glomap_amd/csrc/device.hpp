// device.hpp — device-side building blocks: wave/block reductions with fixed-order partial
// slots, SO(3) in quaternion form.  gfx950: wavefront = 64 lanes (hard-coded).
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"

namespace gsfm {

// ---- wave / block reductions ----------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// The same sum through the DPP network (row shifts inside the 16-lane rows, then the two row broadcasts): VALU moves
// instead of twelve LDS permutes per double — for kernels that reduce many values per thread.  Valid in lane 63.
// (Other summation tree than wave_sum: not interchangeable where bit-identical partials are compared.)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_move_f64<0x111, 0xf, 0xf>(v);  // row_shr:1
  v += dpp_move_f64<0x112, 0xf, 0xf>(v);  // row_shr:2
  v += dpp_move_f64<0x114, 0xf, 0xe>(v);  // row_shr:4
  v += dpp_move_f64<0x118, 0xf, 0xc>(v);  // row_shr:8   -> lane 15 of every row holds the row's sum
  v += dpp_move_f64<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_move_f64<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2 and 3
  return v;  // valid in lane 63
}
// Block-wide sum of K values per thread through wave_sum_dpp; result valid in thread 0.  kBlock threads (4 waves).
template <int K>
__device__ __forceinline__ void block_sum_dpp(double (&v)[K], double* smem /* >= 4*K doubles */) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum_dpp(v[k]);
  __syncthreads();  // protect smem reuse across consecutive calls
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < K; ++k) smem[wave * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += smem[w * K + k];
      v[k] = s;
    }
  }
}

// Sum over the `W`-lane aligned group the lane belongs to; result valid in every lane of the group.
template <int W>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum of K values per thread; result valid in thread 0. kBlock threads (4 waves).
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* smem /* >= 4*K doubles */) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();  // protect smem reuse across consecutive calls
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) smem[wave * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += smem[w * K + k];
      v[k] = s;
    }
  }
}

// Per-block partial slots: part[block * K + k].  The producer kernel writes its slot; every block
// of the consumer kernel re-reduces all `nblocks` slots in the same fixed order, so all blocks
// (and all runs) obtain bit-identical totals.  Result broadcast to all threads of the block.
template <int K>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ part, int nblocks,
                                                double (&out)[K], double* smem /* >= 4*K + K */) {
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] += part[b * K + k];
  }
  block_sum<K>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) smem[4 * K + k] = acc[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = smem[4 * K + k];
  __syncthreads();
}

// ---- SO(3) (quaternion form; q = (w,x,y,z), R(q) x, q_a * q_b <-> R_a R_b) -------------------
struct Quat {
  double w, x, y, z;
};

__device__ __forceinline__ Quat qmul(const Quat& a, const Quat& b) {
  return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
              a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Quat qconj(const Quat& a) { return Quat{a.w, -a.x, -a.y, -a.z}; }

// Exp: angle-axis -> quaternion.  Reference AngleAxisToRotation (glomap/math/rigid3d.cc:45-63):
// Rodrigues for |a| > EPS (1e-12, glomap/types.h:14), first-order I + [a]x otherwise, whose
// quaternion is (1, a/2) to the same order.
__device__ __forceinline__ Quat aa_to_quat(double ax, double ay, double az) {
  const double th2 = ax * ax + ay * ay + az * az;
  const double th = sqrt(th2);
  if (th > 1e-12) {
    double s, c;
    sincos(0.5 * th, &s, &c);
    const double k = s / th;
    return Quat{c, k * ax, k * ay, k * az};
  }
  return Quat{1.0, 0.5 * ax, 0.5 * ay, 0.5 * az};
}

// Log: quaternion -> angle-axis with angle in [0, pi].  Reference RotationToAngleAxis
// (glomap/math/rigid3d.cc:39-43) = Eigen::AngleAxisd(R): angle = 2 atan2(|v|, |w|), axis = v/|v|
// with the sign of w folded into the axis.  Scale-invariant in q, so no renormalisation needed.
__device__ __forceinline__ void quat_to_aa(const Quat& q, double& ax, double& ay, double& az) {
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (n > 0.0) {
    const double ang = 2.0 * atan2(n, fabs(q.w));
    const double k = (q.w < 0.0 ? -ang : ang) / n;
    ax = k * q.x;
    ay = k * q.y;
    az = k * q.z;
  } else {
    ax = ay = az = 0.0;
  }
}

__device__ __forceinline__ Quat load_quat(const double* __restrict__ p) {
  // 32-byte aligned by construction ([.][4] doubles): two 16-byte loads
  const double2 a = *reinterpret_cast<const double2*>(p);
  const double2 b = *reinterpret_cast<const double2*>(p + 2);
  return Quat{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ void store_quat(double* __restrict__ p, const Quat& q) {
  *reinterpret_cast<double2*>(p) = make_double2(q.w, q.x);
  *reinterpret_cast<double2*>(p + 2) = make_double2(q.y, q.z);
}

}  // namespace gsfm
