// cg.hpp — preconditioned conjugate gradients on the reduced camera system (GP: 3N unknowns,
// BA: 6N pose + 8K intrinsics unknowns), single-reduction (Chronopoulos-Gear) form.
//
// Replaces the sparse Cholesky of the Schur complement that Ceres' SPARSE_SCHUR performs for the
// reference (global_positioning.cc:553, bundle_adjustment.cc:95).  The Schur complement is never
// formed: the solver supplies `apply` = w <- (S + D) z as sweeps over its observations.
//
// Why the single-reduction form: classic PCG needs the global scalar p.Ap BEFORE the vector
// update and r.z AFTER it, i.e. two dependent grid-wide reductions per iteration.  With
//     w = A z,  gamma = r.z,  delta = z.w,
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old),
//     p = z + beta p,  s = w + beta s,  x += alpha p,  r -= alpha s,  z = M^-1 r
// both dot products are available right after the mat-vec, so one iteration is
//     [apply kernels]  ->  k_cg_update
// with every scalar resident on the device: partial sums go to fixed per-block slots and each
// consumer block re-reduces them in a fixed order (deterministic; no atomics, no host round trip).
//
// Convergence (|r|^2 <= tol^2 |b|^2) is tested by cg_converged(), which the solver calls at the top
// of its first apply kernel; it raises status->done and every later kernel of the stream returns
// at once, so the host only polls the status every few iterations.
//
// Multi-rank: `apply` produces this rank's share of w and of delta; the solver all-reduces
// w[0..n] (delta travels in w[n]) over RCCL, then k_cg_update runs redundantly on every rank.
#pragma once

#include "device.hpp"

namespace gsfm {

constexpr int kCgMaxBlocks = 512;  // grid cap of the vector kernels (k_cg_init / k_cg_update)
constexpr int kMaxApplySlots = 4096;  // cap of the delta partial slots one apply kernel may write

struct CgStatus {
  int done;
  int iters;
  int bad;  // non-finite or non-positive curvature encountered
  int pad;
  double bb;  // |b|^2
  double rr;  // |r|^2 of the last tested iterate
  double gamma;  // single-workgroup mode: r.z of the current iterate
};
struct CgScal {
  double gamma;  // r.z of the previous iteration
  double alpha;  // step length of the previous iteration
};

// Device view.  Unknown layout: [PB per camera | 8 per intrinsics block]; block-Jacobi blocks are
// the PB x PB camera blocks (minv + PB*PB*n) followed by 8 x 8 intrinsics blocks.
struct CgVec {
  int n = 0;   // PB*N + 8*K
  int N = 0, K = 0;
  int nb_update = 1;  // grid of k_cg_init / k_cg_update  (<= kCgMaxBlocks)
  int nb_apply = 0;   // number of delta partial slots written by `apply`
  int delta_in_w = 0; // multi-rank: delta was all-reduced into w[n]
  int single = 0;     // small systems: the vector update runs in ONE 1024-thread workgroup (k_cg_*1) that
                      // also owns every scalar and the convergence flag — no partial slots to re-reduce
  double tol2 = 0.0;  // squared relative tolerance (single mode: tested inside k_cg_update1)
  const double* b = nullptr;
  double *x = nullptr, *r = nullptr, *z = nullptr, *p = nullptr, *s = nullptr;
  double* w = nullptr;         // [n + 2]
  const double* minv = nullptr;
  double* vpart = nullptr;     // [2][kCgMaxBlocks][2]  (r.z, r.r) partials, double-buffered by iteration parity
  double* dpart = nullptr;     // [nb_apply] delta partials
  CgScal* scal = nullptr;      // [2]
  CgStatus* st = nullptr;
  // optional mirror of the camera part of z inside a solver-owned gather record:
  // zmir[b * zmir_stride + zmir_off + i] = z[PB * b + i]  (lets phase A fetch camera constants and z
  // of a camera with the same 16-byte gathers)
  double* zmir = nullptr;
  int zmir_stride = 0, zmir_off = 0;
  // joint preconditioner (BA with one intrinsics block per camera): camera n and intrinsics block
  // joint_map[n] form ONE (PB + 8) x (PB + 8) block-Jacobi block; minv_joint is stored transposed,
  // minv_joint[(i * BJ + j) * N + n], so that one thread per camera reads it fully coalesced.
  const int* joint_map = nullptr;
  const double* minv_joint = nullptr;
  // joint mode only: per-camera gather record of phase A, zrec[n * zrec_stride + (0..PB-1 | PB + column)]
  // = z of the pose and of the FREE intrinsics of camera n in stored-column order;
  // zrec_slot[8 k + param] = stored column of that parameter of intrinsics block k, or -1
  double* zrec = nullptr;
  int zrec_stride = 0;
  const signed char* zrec_slot = nullptr;
};

template <int BS>
__device__ __forceinline__ void cg_block_init(const CgVec& v, long o, const double* __restrict__ m, double& rz,
                                              double& rr, double* __restrict__ mir = nullptr) {
  double rb[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) rb[i] = v.b[o + i];
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    double zi = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) zi += m[i * BS + j] * rb[j];
    v.x[o + i] = 0.0;
    v.r[o + i] = rb[i];
    v.z[o + i] = zi;
    if (mir) mir[i] = zi;
    v.p[o + i] = 0.0;
    v.s[o + i] = 0.0;
    rz += rb[i] * zi;
    rr += rb[i] * rb[i];
  }
}

// x = 0, r = b, z = M^-1 b, p = s = 0; partials of (r.z, r.r) into parity slot 0.
template <int PB, bool HAS_INTR>
static __global__ void __launch_bounds__(kBlock) k_cg_init(CgVec v) {
  __shared__ double smem[4 * 2];
  double acc[2] = {0.0, 0.0};
  const int nblk = v.N + (HAS_INTR ? v.K : 0);
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += gridDim.x * blockDim.x) {
    if (b < v.N) {
      cg_block_init<PB>(v, (long)PB * b, v.minv + (long)PB * PB * b, acc[0], acc[1],
                        v.zmir ? v.zmir + (long)b * v.zmir_stride + v.zmir_off : nullptr);
    } else if constexpr (HAS_INTR) {
      const int k = b - v.N;
      cg_block_init<8>(v, (long)PB * v.N + 8L * k, v.minv + (long)PB * PB * v.N + 64L * k, acc[0], acc[1]);
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    v.vpart[blockIdx.x * 2] = acc[0];
    v.vpart[blockIdx.x * 2 + 1] = acc[1];
    if (blockIdx.x == 0) {
      v.st->done = 0;
      v.st->iters = 0;
      v.st->bad = 0;
      v.st->bb = 0.0;
      v.st->rr = 0.0;
      v.scal[0].gamma = 0.0;
      v.scal[0].alpha = 0.0;
    }
  }
}

// To be called by ALL threads of EVERY block at the top of the first apply kernel of iteration
// `it` (kBlock threads).  Returns true when the solve is finished (the caller returns).
__device__ __forceinline__ bool cg_converged(const CgVec& v, int it, double tol2, double* smem /* >= 4*2+2 */) {
  if (v.single) return v.st->done != 0;  // k_cg_init1 / k_cg_update1 already tested |r| and raised `done`
  double t[2] = {0.0, 0.0};
  {
    const double* vp = v.vpart + (size_t)(it & 1) * kCgMaxBlocks * 2;
    for (int b = threadIdx.x; b < v.nb_update; b += blockDim.x) {  // issued before the `done` round trip
      t[0] += vp[2 * b];
      t[1] += vp[2 * b + 1];
    }
  }
  const int done0 = v.st->done;
  if (done0) return true;
  block_sum<2>(t, smem);
  if (threadIdx.x == 0) {
    smem[8] = t[0];
    smem[9] = t[1];
  }
  __syncthreads();
  t[0] = smem[8];
  t[1] = smem[9];
  __syncthreads();
  const double bb = it == 0 ? t[1] : v.st->bb;
  const bool finite = isfinite(t[0]) && isfinite(t[1]);
  const bool done = !finite || t[1] <= tol2 * bb;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (it == 0) v.st->bb = bb;
    v.st->rr = t[1];
    v.st->iters = it;
    if (!finite) v.st->bad = 1;
    if (done) v.st->done = 1;
  }
  return done;
}

template <int BS>
__device__ __forceinline__ void cg_block_update(const CgVec& v, long o, const double* __restrict__ m, double alpha,
                                                double beta, double& rz, double& rr,
                                                double* __restrict__ mir = nullptr) {
  double rn[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    const double pi = v.z[o + i] + beta * v.p[o + i];
    const double si = v.w[o + i] + beta * v.s[o + i];
    v.p[o + i] = pi;
    v.s[o + i] = si;
    v.x[o + i] += alpha * pi;
    rn[i] = v.r[o + i] - alpha * si;
    v.r[o + i] = rn[i];
    rr += rn[i] * rn[i];
  }
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    double zi = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) zi += m[i * BS + j] * rn[j];
    v.z[o + i] = zi;
    if (mir) mir[i] = zi;
    rz += rn[i] * zi;
  }
}

// One CG iteration given w = A z (complete) and the delta partials.
template <int PB, bool HAS_INTR>
static __global__ void __launch_bounds__(kBlock) k_cg_update(CgVec v, int it) {
  __shared__ double smem[4 * 3 + 3];
  // one pass over both sets of partial slots (r.z | r.r of the previous update, delta of this apply);
  // the loads are issued before the `done` test so that they overlap its round trip
  double t3[3] = {0.0, 0.0, 0.0};
  {
    const double* vp = v.vpart + (size_t)(it & 1) * kCgMaxBlocks * 2;
    for (int b = threadIdx.x; b < v.nb_update; b += blockDim.x) {
      t3[0] += vp[2 * b];
      t3[1] += vp[2 * b + 1];
    }
    if (!v.delta_in_w)
      for (int b = threadIdx.x; b < v.nb_apply; b += blockDim.x) t3[2] += v.dpart[b];
  }
  const CgScal prev = v.scal[it & 1];
  const int done = v.st->done;
  if (done) return;
  __shared__ double sbc[3];
  block_sum<3>(t3, smem);
  if (threadIdx.x == 0) {
    sbc[0] = t3[0];
    sbc[1] = t3[1];
    sbc[2] = t3[2];
  }
  __syncthreads();
  const double gamma = sbc[0];
  const double delta = v.delta_in_w ? v.w[v.n] : sbc[2];
  __syncthreads();
  double beta = 0.0, denom = delta;
  if (it > 0) {
    beta = prev.gamma > 0.0 ? gamma / prev.gamma : 0.0;
    denom = delta - beta * gamma / prev.alpha;
  }
  const bool ok = isfinite(denom) && denom > 0.0 && isfinite(gamma) && gamma >= 0.0;
  const double alpha = ok ? gamma / denom : 0.0;
  if (!ok) beta = 0.0;
  double acc[2] = {0.0, 0.0};
  const int nblk = v.N + (HAS_INTR ? v.K : 0);
  if (ok) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += gridDim.x * blockDim.x) {
      if (b < v.N) {
        cg_block_update<PB>(v, (long)PB * b, v.minv + (long)PB * PB * b, alpha, beta, acc[0], acc[1],
                            v.zmir ? v.zmir + (long)b * v.zmir_stride + v.zmir_off : nullptr);
      } else if constexpr (HAS_INTR) {
        const int k = b - v.N;
        cg_block_update<8>(v, (long)PB * v.N + 8L * k, v.minv + (long)PB * PB * v.N + 64L * k, alpha, beta, acc[0],
                           acc[1]);
      }
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    double* out = v.vpart + (size_t)((it + 1) & 1) * kCgMaxBlocks * 2;
    out[blockIdx.x * 2] = acc[0];
    out[blockIdx.x * 2 + 1] = acc[1];
    if (blockIdx.x == 0) {
      v.scal[(it + 1) & 1].gamma = gamma;
      v.scal[(it + 1) & 1].alpha = alpha;
      if (!ok) {
        // gamma == 0 with a zero residual is plain convergence, anything else is a breakdown
        if (!(gamma == 0.0 && isfinite(delta))) v.st->bad = 1;
        v.st->done = 1;
        v.st->iters = it;
      }
    }
  }
}



// ---- single-workgroup vector kernels (small systems: RA node vectors, GP with few thousand cameras) ----
constexpr int kCgSingleThreads = 1024;
constexpr int kCgSingleMaxBlocks = 1024;  // block-Jacobi blocks handled by one workgroup (one CU's bandwidth beyond that)

template <int K>
__device__ __forceinline__ void block_sum_1024(double (&v)[K], double* smem /* >= 16 K + K */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) smem[wave * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double t = 0.0;
      for (int w = 0; w < nw; ++w) t += smem[w * K + k];
      smem[16 * K + k] = t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = smem[16 * K + k];
  __syncthreads();
}

template <int PB>
static __global__ void __launch_bounds__(kCgSingleThreads) k_cg_init1(CgVec v) {
  __shared__ double smem[16 * 2 + 2];
  double acc[2] = {0.0, 0.0};
  for (int b = threadIdx.x; b < v.N; b += blockDim.x)
    cg_block_init<PB>(v, (long)PB * b, v.minv + (long)PB * PB * b, acc[0], acc[1],
                      v.zmir ? v.zmir + (long)b * v.zmir_stride + v.zmir_off : nullptr);
  block_sum_1024<2>(acc, smem);
  if (threadIdx.x == 0) {
    v.st->done = (acc[1] == 0.0 || !isfinite(acc[1])) ? 1 : 0;
    v.st->iters = 0;
    v.st->bad = isfinite(acc[1]) ? 0 : 1;
    v.st->bb = acc[1];
    v.st->rr = acc[1];
    v.st->gamma = acc[0];
    v.scal[0].gamma = 0.0;
    v.scal[0].alpha = 0.0;
  }
}

template <int PB>
static __global__ void __launch_bounds__(kCgSingleThreads) k_cg_update1(CgVec v, int it) {
  __shared__ double smem[16 * 2 + 2];
  if (v.st->done) return;
  double d[1] = {0.0};
  if (v.delta_in_w) {
    d[0] = v.w[v.n];
  } else {
    for (int b = threadIdx.x; b < v.nb_apply; b += blockDim.x) d[0] += v.dpart[b];
    block_sum_1024<1>(d, smem);
  }
  const double delta = d[0];
  const double gamma = v.st->gamma;
  const double bb = v.st->bb;
  const CgScal prev = v.scal[it & 1];
  __syncthreads();  // everyone has read the scalars thread 0 rewrites below
  double beta = 0.0, denom = delta;
  if (it > 0) {
    beta = prev.gamma > 0.0 ? gamma / prev.gamma : 0.0;
    denom = delta - beta * gamma / prev.alpha;
  }
  const bool ok = isfinite(denom) && denom > 0.0 && isfinite(gamma) && gamma >= 0.0;
  const double alpha = ok ? gamma / denom : 0.0;
  double acc[2] = {0.0, 0.0};
  if (ok) {
    for (int b = threadIdx.x; b < v.N; b += blockDim.x)
      cg_block_update<PB>(v, (long)PB * b, v.minv + (long)PB * PB * b, alpha, beta, acc[0], acc[1],
                          v.zmir ? v.zmir + (long)b * v.zmir_stride + v.zmir_off : nullptr);
  }
  block_sum_1024<2>(acc, smem);
  if (threadIdx.x == 0) {
    v.scal[(it + 1) & 1].gamma = gamma;
    v.scal[(it + 1) & 1].alpha = alpha;
    if (!ok) {
      if (!(gamma == 0.0 && isfinite(delta))) v.st->bad = 1;
      v.st->done = 1;
      v.st->iters = it;
    } else {
      const bool finite = isfinite(acc[0]) && isfinite(acc[1]);
      v.st->gamma = acc[0];
      v.st->rr = acc[1];
      v.st->iters = it + 1;
      if (!finite) v.st->bad = 1;
      if (!finite || acc[1] <= v.tol2 * bb) v.st->done = 1;
    }
  }
}

// ---- joint (pose + intrinsics) blocks ----------------------------------------------------------
template <int PB>
__device__ __forceinline__ long cg_joint_index(const CgVec& v, int n, int i) {
  return i < PB ? (long)PB * n + i : (long)PB * v.N + 8L * v.joint_map[n] + (i - PB);
}
// Joint kernels: 16 lanes per camera (PB + 8 <= 16 active), 16 cameras per workgroup.  Lane (c, i)
// owns element i of camera c's joint block; the preconditioner row product reads the block's
// residual from LDS.  minv_joint[(i * BJ + j) * N + n]: for a fixed (i, j) the 16 cameras of a
// workgroup read 16 consecutive doubles.
constexpr int kJointCams = kBlock / 16;

template <int PB>
__device__ __forceinline__ void cg_joint_mirror(const CgVec& v, int n, int i, double zi) {
  if (v.zrec == nullptr) return;
  if (i < PB) {
    v.zrec[(long)n * v.zrec_stride + i] = zi;
  } else {
    const int slot = v.zrec_slot[8 * v.joint_map[n] + (i - PB)];
    if (slot >= 0) v.zrec[(long)n * v.zrec_stride + PB + slot] = zi;
  }
}

template <int PB>
static __global__ void __launch_bounds__(kBlock) k_cg_init_joint(CgVec v) {
  constexpr int BJ = PB + 8;
  __shared__ double smem[4 * 2];
  __shared__ double sr[kJointCams][16];
  const int c = threadIdx.x >> 4, i = threadIdx.x & 15;
  double acc[2] = {0.0, 0.0};
  for (int n0 = blockIdx.x * kJointCams; n0 < v.N; n0 += gridDim.x * kJointCams) {
    const int n = n0 + c;
    const bool act = n < v.N && i < BJ;
    long o = 0;
    double ri = 0.0;
    if (act) {
      o = cg_joint_index<PB>(v, n, i);
      ri = v.b[o];
    }
    __syncthreads();
    sr[c][i] = ri;
    __syncthreads();
    if (act) {
      const double* m = v.minv_joint + n;
      double zi = 0.0;
#pragma unroll
      for (int j = 0; j < BJ; ++j) zi += m[(size_t)(i * BJ + j) * v.N] * sr[c][j];
      v.x[o] = 0.0;
      v.r[o] = ri;
      v.z[o] = zi;
      v.p[o] = 0.0;
      v.s[o] = 0.0;
      cg_joint_mirror<PB>(v, n, i, zi);
      acc[0] += ri * zi;
      acc[1] += ri * ri;
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    v.vpart[blockIdx.x * 2] = acc[0];
    v.vpart[blockIdx.x * 2 + 1] = acc[1];
    if (blockIdx.x == 0) {
      v.st->done = 0;
      v.st->iters = 0;
      v.st->bad = 0;
      v.st->bb = 0.0;
      v.st->rr = 0.0;
      v.scal[0].gamma = 0.0;
      v.scal[0].alpha = 0.0;
    }
  }
}

template <int PB>
static __global__ void __launch_bounds__(kBlock) k_cg_update_joint(CgVec v, int it) {
  constexpr int BJ = PB + 8;
  __shared__ double smem[4 * 2 + 2];
  __shared__ double sr[kJointCams][16];
  if (v.st->done) return;
  double g[2];
  reduce_partials<2>(v.vpart + (size_t)(it & 1) * kCgMaxBlocks * 2, v.nb_update, g, smem);
  double delta;
  if (v.delta_in_w) {
    delta = v.w[v.n];
  } else {
    double d[1];
    reduce_partials<1>(v.dpart, v.nb_apply, d, smem);
    delta = d[0];
  }
  const double gamma = g[0];
  const CgScal prev = v.scal[it & 1];
  double beta = 0.0, denom = delta;
  if (it > 0) {
    beta = prev.gamma > 0.0 ? gamma / prev.gamma : 0.0;
    denom = delta - beta * gamma / prev.alpha;
  }
  const bool ok = isfinite(denom) && denom > 0.0 && isfinite(gamma) && gamma >= 0.0;
  const double alpha = ok ? gamma / denom : 0.0;
  if (!ok) beta = 0.0;
  double acc[2] = {0.0, 0.0};
  const int c = threadIdx.x >> 4, i = threadIdx.x & 15;
  if (ok) {
    for (int n0 = blockIdx.x * kJointCams; n0 < v.N; n0 += gridDim.x * kJointCams) {
      const int n = n0 + c;
      const bool act = n < v.N && i < BJ;
      long o = 0;
      double ri = 0.0;
      if (act) {
        o = cg_joint_index<PB>(v, n, i);
        const double pi = v.z[o] + beta * v.p[o];
        const double si = v.w[o] + beta * v.s[o];
        v.p[o] = pi;
        v.s[o] = si;
        v.x[o] += alpha * pi;
        ri = v.r[o] - alpha * si;
        v.r[o] = ri;
        acc[1] += ri * ri;
      }
      __syncthreads();
      sr[c][i] = ri;
      __syncthreads();
      if (act) {
        const double* m = v.minv_joint + n;
        double zi = 0.0;
#pragma unroll
        for (int j = 0; j < BJ; ++j) zi += m[(size_t)(i * BJ + j) * v.N] * sr[c][j];
        v.z[o] = zi;
        cg_joint_mirror<PB>(v, n, i, zi);
        acc[0] += ri * zi;
      }
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    double* out = v.vpart + (size_t)((it + 1) & 1) * kCgMaxBlocks * 2;
    out[blockIdx.x * 2] = acc[0];
    out[blockIdx.x * 2 + 1] = acc[1];
    if (blockIdx.x == 0) {
      v.scal[(it + 1) & 1].gamma = gamma;
      v.scal[(it + 1) & 1].alpha = alpha;
      if (!ok) {
        if (!(gamma == 0.0 && isfinite(delta))) v.st->bad = 1;
        v.st->done = 1;
        v.st->iters = it;
      }
    }
  }
}

// Single-block reduction of the delta partials into w[n] (multi-rank: w[0..n] is all-reduced next).
static __global__ void __launch_bounds__(kBlock) k_cg_delta_to_w(CgVec v) {
  __shared__ double smem[4 + 1];
  if (v.st->done) return;
  double d[1];
  reduce_partials<1>(v.dpart, v.nb_apply, d, smem);
  if (threadIdx.x == 0) v.w[v.n] = d[0];
}

// ---- deflation of known near-null modes (EXPERIMENT, off unless a solver passes a CgDeflation) ---------------------
// The block-Jacobi-preconditioned reduced systems of BA / GP have a tight spectrum plus the similarity gauge of the
// scene (CPU evidence tools/exp_deflation.py; executable specification: pcg(..., defl) in
// oracle/csrc/orc_lm.hpp).  Deflated PCG (Saad, Yeung, Erhel, Guyomarc'h 2000) on the span of k given modes W:
//     E = W^T A W,   x = W E^-1 W^T b + x',   A x' = b2 := b - A W E^-1 W^T b,
//     every preconditioned residual is projected:  z <- z - W E^-1 (A W)^T z   (r.z is patched by -(W^T r).(E^-1 (AW)^T z))
// Same system, same stopping rule (relative to |b2| <= |b|).  Written at the end of round 2 without a GPU to run it on:
// nothing here executes unless GSFM_DEFLATE is set (gp.hip / ba.hip), single rank, multi-block vector kernels only.
constexpr int kCgMaxModes = 8;
constexpr int kCgdBlocks = 128;  // grid of the dot-product kernel

struct CgDeflation {
  int k = 0;                   // modes in use (<= kCgMaxModes)
  const double* W = nullptr;   // [k][n] the modes (zero in constant components)
  double* AW = nullptr;        // [k][n]
  double* b2 = nullptr;        // [n]   deflated right-hand side
  double* part = nullptr;      // [kCgdBlocks][2 * kCgMaxModes] partial dot products
  double* small = nullptr;     // [64] E^-1 | [8] y0 = E^-1 W^T b | [1] ok (1.0 / 0.0)
};

// the element list of the unknown vector with its gather mirrors: f(o, n, i) for every unknown o; (n, i) = camera and
// index inside its joint block when JOINT, else (camera block or -1, index)
template <int PB, bool JOINT, typename F>
__device__ __forceinline__ void cgd_for_each(const CgVec& v, F f) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  if constexpr (JOINT) {
    for (long e = tid; e < (long)v.N * 16; e += nth) {
      const int n = (int)(e >> 4), i = (int)(e & 15);
      if (i < PB + 8) f(cg_joint_index<PB>(v, n, i), n, i);
    }
  } else {
    for (long o = tid; o < (long)v.n; o += nth) {
      const bool cam = o < (long)PB * v.N;
      f(o, cam ? (int)(o / PB) : -1, cam ? (int)(o % PB) : 0);
    }
  }
}
template <int PB, bool JOINT>
__device__ __forceinline__ void cgd_store_z(const CgVec& v, long o, int n, int i, double zi) {
  v.z[o] = zi;
  if constexpr (JOINT) {
    cg_joint_mirror<PB>(v, n, i, zi);
  } else {
    if (v.zmir && n >= 0) v.zmir[(long)n * v.zmir_stride + v.zmir_off + i] = zi;
  }
}

// z := vec (with mirrors): the input of one operator application
template <int PB, bool JOINT>
static __global__ void __launch_bounds__(kBlock) k_cgd_set_z(CgVec v, const double* __restrict__ vec) {
  cgd_for_each<PB, JOINT>(v, [&](long o, int n, int i) { cgd_store_z<PB, JOINT>(v, o, n, i, vec[o]); });
}
static __global__ void __launch_bounds__(kBlock) k_cgd_copy(long n, const double* __restrict__ src, double* __restrict__ dst) {
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) dst[o] = src[o];
}

// E = W^T A W (symmetrised), E^-1 by Gauss-Jordan with a positivity test, y0 = E^-1 W^T b, b2 = b - A W y0.  One
// workgroup: k^2 + k dot products over n <= a few 10^5 doubles, once per solve.
static __global__ void __launch_bounds__(kCgSingleThreads) k_cgd_gram(CgVec v, CgDeflation d) {
  __shared__ double smem[16 * 1 + 1];
  __shared__ double E[kCgMaxModes][kCgMaxModes + 1];  // last column: W^T b
  __shared__ double sy[kCgMaxModes];
  __shared__ int sok;
  const int k = d.k;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j <= k; ++j) {
      const double* a = d.W + (size_t)i * v.n;
      const double* c = j < k ? d.AW + (size_t)j * v.n : v.b;
      double t[1] = {0.0};
      for (long o = threadIdx.x; o < (long)v.n; o += blockDim.x) t[0] += a[o] * c[o];
      block_sum_1024<1>(t, smem);
      if (threadIdx.x == 0) E[i][j] = t[0];
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    double A[kCgMaxModes][2 * kCgMaxModes];
    bool ok = true;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j) {
        A[i][j] = 0.5 * (E[i][j] + E[j][i]);
        A[i][k + j] = i == j ? 1.0 : 0.0;
      }
    for (int c = 0; c < k && ok; ++c) {  // SPD: the pivots are the diagonal
      const double piv = A[c][c];
      if (!(piv > 0.0) || !isfinite(piv)) {
        ok = false;
        break;
      }
      const double ip = 1.0 / piv;
      for (int j = 0; j < 2 * k; ++j) A[c][j] *= ip;
      for (int r = 0; r < k; ++r) {
        if (r == c) continue;
        const double f = A[r][c];
        for (int j = 0; j < 2 * k; ++j) A[r][j] -= f * A[c][j];
      }
    }
    for (int i = 0; i < k; ++i) {
      double y = 0.0;
      for (int j = 0; j < k; ++j) {
        const double e = ok ? A[i][k + j] : 0.0;
        d.small[i * kCgMaxModes + j] = e;
        y += e * E[j][k];
      }
      if (!isfinite(y)) ok = false;
      sy[i] = y;
    }
    if (!ok)
      for (int i = 0; i < k; ++i) sy[i] = 0.0;
    for (int i = 0; i < k; ++i) d.small[64 + i] = sy[i];
    d.small[72] = ok ? 1.0 : 0.0;
    sok = ok ? 1 : 0;
  }
  __syncthreads();
  for (long o = threadIdx.x; o < (long)v.n; o += blockDim.x) {
    double t = v.b[o];
    if (sok)
      for (int j = 0; j < k; ++j) t -= sy[j] * d.AW[(size_t)j * v.n + o];
    d.b2[o] = t;
  }
}

// partial dot products  c_j = (A W_j) . z,  dd_j = W_j . r   of the current iterate
static __global__ void __launch_bounds__(kBlock) k_cgd_dots(CgVec v, CgDeflation d) {
  __shared__ double smem[4 * 2 * kCgMaxModes];
  if (v.st->done || d.small[72] == 0.0) return;
  double acc[2 * kCgMaxModes];
#pragma unroll
  for (int j = 0; j < 2 * kCgMaxModes; ++j) acc[j] = 0.0;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < (long)v.n; o += (long)gridDim.x * blockDim.x) {
    const double zo = v.z[o], ro = v.r[o];
#pragma unroll
    for (int j = 0; j < kCgMaxModes; ++j)
      if (j < d.k) {
        acc[j] += d.AW[(size_t)j * v.n + o] * zo;
        acc[kCgMaxModes + j] += d.W[(size_t)j * v.n + o] * ro;
      }
  }
  block_sum<2 * kCgMaxModes>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 2 * kCgMaxModes; ++j) d.part[blockIdx.x * 2 * kCgMaxModes + j] = acc[j];
  }
}

// z <- z - W E^-1 c (vector and mirrors); block 0 patches the r.z partial of parity slot (it + 1) & 1 by -dd . (E^-1 c)
template <int PB, bool JOINT>
static __global__ void __launch_bounds__(kBlock) k_cgd_project(CgVec v, CgDeflation d, int it, int nparts) {
  __shared__ double smem[5 * 2 * kCgMaxModes];
  __shared__ double sy[kCgMaxModes];
  if (v.st->done || d.small[72] == 0.0) return;
  double cd[2 * kCgMaxModes];
  reduce_partials<2 * kCgMaxModes>(d.part, nparts, cd, smem);
  if (threadIdx.x == 0) {
    double corr = 0.0;
    for (int i = 0; i < d.k; ++i) {
      double y = 0.0;
      for (int j = 0; j < d.k; ++j) y += d.small[i * kCgMaxModes + j] * cd[j];
      sy[i] = y;
      corr += y * cd[kCgMaxModes + i];
    }
    if (blockIdx.x == 0) v.vpart[(size_t)((it + 1) & 1) * kCgMaxBlocks * 2] -= corr;
  }
  __syncthreads();
  cgd_for_each<PB, JOINT>(v, [&](long o, int n, int i) {
    double zi = v.z[o];
    for (int j = 0; j < d.k; ++j) zi -= sy[j] * d.W[(size_t)j * v.n + o];
    cgd_store_z<PB, JOINT>(v, o, n, i, zi);
  });
}

// x <- x + W y0
static __global__ void __launch_bounds__(kBlock) k_cgd_finish(CgVec v, CgDeflation d) {
  if (d.small[72] == 0.0) return;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < (long)v.n; o += (long)gridDim.x * blockDim.x) {
    double xo = v.x[o];
    for (int j = 0; j < d.k; ++j) xo += d.small[64 + j] * d.W[(size_t)j * v.n + o];
    v.x[o] = xo;
  }
}

// Host driver.  `apply(it)` must enqueue the kernels computing w = A z and the delta partials
// (its first kernel calls cg_converged); it is also responsible for timing its dominant kernel.
template <int PB, bool HAS_INTR, typename Apply>
inline long cg_solve(gsfm_ctx* ctx, CgVec& v, double tol, int max_iter, Apply&& apply, const CgDeflation* defl = nullptr) {
  hipStream_t s = ctx->stream;
  const bool multi = ctx->comm.world > 1;
  v.delta_in_w = multi ? 1 : 0;
  const bool joint = HAS_INTR && v.joint_map != nullptr;
  v.tol2 = tol * tol;
  v.single = (!HAS_INTR && v.N <= kCgSingleMaxBlocks) ? 1 : 0;
  auto init = [&]() {
    if constexpr (HAS_INTR) {
      if (joint) hipLaunchKernelGGL((k_cg_init_joint<PB>), dim3(v.nb_update), dim3(kBlock), 0, s, v);
    } else {
      if (v.single) hipLaunchKernelGGL((k_cg_init1<PB>), dim3(1), dim3(kCgSingleThreads), 0, s, v);
    }
    if (!joint && !v.single) hipLaunchKernelGGL((k_cg_init<PB, HAS_INTR>), dim3(v.nb_update), dim3(kBlock), 0, s, v);
  };
  init();
  // experiment (see CgDeflation): only the multi-block vector kernels on one rank
  const bool deflate = defl != nullptr && defl->k > 0 && !multi && !v.single;
  const double* b_caller = v.b;
  const int gdot = std::min(kCgdBlocks, grid_for((size_t)v.n, kBlock));
  const int gvec = std::min(256, grid_for((size_t)std::max((long)v.n, (long)v.N * 16), kBlock));
  auto project = [&](int it) {
    hipLaunchKernelGGL(k_cgd_dots, dim3(gdot), dim3(kBlock), 0, s, v, *defl);
    if (joint)
      hipLaunchKernelGGL((k_cgd_project<PB, true>), dim3(gvec), dim3(kBlock), 0, s, v, *defl, it, gdot);
    else
      hipLaunchKernelGGL((k_cgd_project<PB, false>), dim3(gvec), dim3(kBlock), 0, s, v, *defl, it, gdot);
  };
  if (deflate) {
    for (int j = 0; j < defl->k; ++j) {  // A W_j: the operator reads z (and its mirrors), writes w
      if (joint)
        hipLaunchKernelGGL((k_cgd_set_z<PB, true>), dim3(gvec), dim3(kBlock), 0, s, v, defl->W + (size_t)j * v.n);
      else
        hipLaunchKernelGGL((k_cgd_set_z<PB, false>), dim3(gvec), dim3(kBlock), 0, s, v, defl->W + (size_t)j * v.n);
      apply(0);
      hipLaunchKernelGGL(k_cgd_copy, dim3(gdot), dim3(kBlock), 0, s, (long)v.n, (const double*)v.w, defl->AW + (size_t)j * v.n);
    }
    hipLaunchKernelGGL(k_cgd_gram, dim3(1), dim3(kCgSingleThreads), 0, s, v, *defl);
    v.b = defl->b2;
    init();
    project(-1);
  }
  CgStatus* h = reinterpret_cast<CgStatus*>(ctx->h_pinned + 400);
  const int chunk = 8;
  long iters = max_iter;
  for (int it = 0; it < max_iter; ++it) {
    apply(it);
    if (multi) {
      hipLaunchKernelGGL(k_cg_delta_to_w, dim3(1), dim3(kBlock), 0, s, v);
      allreduce_sum(ctx, v.w, (size_t)v.n + 1);
    }
    if constexpr (HAS_INTR) {
      if (joint) hipLaunchKernelGGL((k_cg_update_joint<PB>), dim3(v.nb_update), dim3(kBlock), 0, s, v, it);
    } else {
      if (v.single) hipLaunchKernelGGL((k_cg_update1<PB>), dim3(1), dim3(kCgSingleThreads), 0, s, v, it);
    }
    if (!joint && !v.single) hipLaunchKernelGGL((k_cg_update<PB, HAS_INTR>), dim3(v.nb_update), dim3(kBlock), 0, s, v, it);
    if (deflate) project(it);
    if ((it + 1) % chunk == 0 || it == max_iter - 1) {
      GSFM_HIP_CHECK(hipMemcpyAsync(h, v.st, sizeof(CgStatus), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipStreamSynchronize(s));
      GSFM_HIP_CHECK(hipGetLastError());
      ctx->prof.harvest();
      if (h->done) {
        iters = h->iters;
        break;
      }
    }
  }
  if (deflate) {
    hipLaunchKernelGGL(k_cgd_finish, dim3(gdot), dim3(kBlock), 0, s, v, *defl);
    v.b = b_caller;
    iters += defl->k;  // the operator applications that formed A W
  }
  return iters;
}

}  // namespace gsfm
