// cgvec.hpp — the vector half of the preconditioned conjugate-gradient solve of the reduced camera
// system (GP: 3N, BA: 6N + intrinsics unknowns).
//
// The reduced system is tiny next to the observation sweep that applies it (N cameras vs M
// observations), so all O(n) vector work of one CG iteration — both dot products, the x/r/z/p
// updates, the block-Jacobi preconditioner and the convergence test — runs in ONE single-workgroup
// kernel (1024 threads, wave-shuffle + LDS reductions): no cross-workgroup reduction, no host round
// trip, deterministic summation order.  Per CG iteration the stream sees exactly two launches:
// the implicit-Schur mat-vec over the observations (y += S p) and k_cg_iter.
//
// With several ranks, y is the only thing exchanged: each rank accumulates the contribution of its
// own tracks, then one RCCL all-reduce (sum) of y; every rank runs k_cg_iter redundantly on the
// replicated vectors, so scalars never need a collective.
#pragma once

#include "device.hpp"

namespace gsfm {

constexpr int kCgThreads = 1024;

struct CgState {
  double rz;  // r.z of the current iterate
  double bb;  // |b|^2
  double rr;  // |r|^2
  int done;
  int iters;
  int bad;    // non-finite or non-positive curvature seen
};

// Variable-size block-Jacobi preconditioner: element i belongs to block elem_blk[i] which starts
// at blk_start[b], has blk_size[b] unknowns and a dense row-major inverse at minv + blk_moff[b].
struct BlockJacobi {
  const int* elem_blk;
  const int* blk_start;
  const int* blk_size;
  const int* blk_moff;
  const double* minv;
};

__device__ __forceinline__ double block_dot_1024(double v, double* smem /* >= 17 doubles */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) s += smem[w];
    smem[16] = s;
  }
  __syncthreads();
  return smem[16];
}

__device__ __forceinline__ double apply_block_jacobi(const BlockJacobi& bj, int i, const double* __restrict__ r) {
  const int b = bj.elem_blk[i];
  const int s0 = bj.blk_start[b], bs = bj.blk_size[b];
  const double* __restrict__ m = bj.minv + bj.blk_moff[b] + (long)(i - s0) * bs;
  double acc = 0.0;
  for (int j = 0; j < bs; ++j) acc += m[j] * r[s0 + j];
  return acc;
}

// x = 0, r = b, z = M^-1 r, p = z, y = y_init_scale * D p  (D = LM damping of the camera block,
// the part of S that is not produced by the observation sweep).
static __global__ void __launch_bounds__(kCgThreads)
    k_cg_init(int n, const double* __restrict__ b, double* __restrict__ x, double* __restrict__ r,
              double* __restrict__ z, double* __restrict__ p, double* __restrict__ y,
              const double* __restrict__ dvec, BlockJacobi bj, CgState* __restrict__ st,
              double y_init_scale) {
  __shared__ double smem[17];
  double bb = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double bi = b[i];
    x[i] = 0.0;
    r[i] = bi;
    bb += bi * bi;
  }
  __syncthreads();
  double rz = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double zi = apply_block_jacobi(bj, i, r);
    z[i] = zi;
    p[i] = zi;
    y[i] = y_init_scale * dvec[i] * zi;
    rz += r[i] * zi;
  }
  bb = block_dot_1024(bb, smem);
  rz = block_dot_1024(rz, smem);
  if (threadIdx.x == 0) {
    st->rz = rz;
    st->bb = bb;
    st->rr = bb;
    st->done = (bb == 0.0) ? 1 : 0;
    st->iters = 0;
    st->bad = 0;
  }
}

// One CG iteration given y = S p (complete, all-reduced):
//   alpha = rz / p.y;  x += alpha p;  r -= alpha y;  z = M^-1 r;  beta = rz' / rz;  p = z + beta p
//   done when |r| <= tol |b|;  y is re-initialised to D p for the next mat-vec.
static __global__ void __launch_bounds__(kCgThreads)
    k_cg_iter(int n, double* __restrict__ y, double* __restrict__ p, double* __restrict__ x,
              double* __restrict__ r, double* __restrict__ z, const double* __restrict__ dvec,
              BlockJacobi bj, CgState* __restrict__ st, double tol2, double y_init_scale) {
  __shared__ double smem[17];
  if (st->done) return;
  double pq = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) pq += p[i] * y[i];
  pq = block_dot_1024(pq, smem);
  const double rz = st->rz;
  const bool ok = pq > 0.0 && isfinite(pq);
  const double alpha = ok ? rz / pq : 0.0;
  double rr = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * y[i];
    r[i] = ri;
    rr += ri * ri;
  }
  __syncthreads();
  double rzn = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double zi = apply_block_jacobi(bj, i, r);
    z[i] = zi;
    rzn += r[i] * zi;
  }
  rr = block_dot_1024(rr, smem);
  rzn = block_dot_1024(rzn, smem);
  const double beta = rz > 0.0 ? rzn / rz : 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double pi = z[i] + beta * p[i];
    p[i] = pi;
    y[i] = y_init_scale * dvec[i] * pi;
  }
  if (threadIdx.x == 0) {
    st->rz = rzn;
    st->rr = rr;
    st->iters += 1;
    if (!ok) {
      st->bad = 1;
      st->done = 1;
    }
    if (rr <= tol2 * st->bb) st->done = 1;
  }
}

}  // namespace gsfm
