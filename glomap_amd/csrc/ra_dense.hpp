// ra_dense.hpp — direct solves of the rotation-averaging Laplacian for small view graphs.
//
// The reference factorises A^T W A with CHOLMOD once for the L1 stage and once per IRLS iteration
// (global_rotation_averaging.cc:491,547-611).  For a view graph of a few thousand frames the
// Laplacian (N x N scalar, thanks to A = B (x) I3) is small enough to treat as DENSE, and an
// iterative solve is the wrong tool: at N = 1000 one PCG iteration is two ~8 us launches, a solve is
// ~50 of them and the L1 stage alone needs 50 solves.  Here the matrix is inverted explicitly by a
// tiled Gauss-Jordan sweep — T = N/32 launches per inversion, each a grid of T x T workgroups doing
// 32x32x32 f64 tile products on the matrix cores (v_mfma_f64_16x16x4_f64) — and every solve is one
// dense (A^-1)[N x N] x rhs[N x 3] product plus one step of iterative refinement against the sparse
// operator.  SPD input => the block pivots are Schur complements of an SPD matrix, no pivoting.
//
// Block Gauss-Jordan step k (P = A_kk^-1), out-of-place between two buffers so that no tile is read after it was
// overwritten, in the symmetric (sweep-operator) form — see k_gj_sweep_step.  The workgroup that produces tile
// (k+1, k+1) also inverts it (in LDS) for the next step.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "device.hpp"

namespace gsfm {

constexpr int kTile = 32;        // tile edge
constexpr int kTileLd = 33;      // LDS leading dimension (bank spread)
constexpr int kDenseMaxN = 2048; // largest view graph solved densely
constexpr int kBlockDenseMaxN = 32768;  // largest view graph preconditioned by dense diagonal blocks of <= kDenseMaxN nodes

using f64x4 = __attribute__((ext_vector_type(4))) double;

// C (32x32) = sign * X (32x32) . Y (32x32) [+ Cin], all tiles in LDS with leading dimension kTileLd.
// 4 waves, one 16x16 output block each, 8 MFMA (16x16x4 f64) per block.
// v_mfma_f64_16x16x4_f64 operand map (MI355X guide §3): A: lane l holds A[l & 15][l >> 4],
// B: lane l holds B[l >> 4][l & 15], C/D: reg i of lane l = C[(l >> 4) + 4 i][l & 15].
__device__ __forceinline__ void tile_mma(const double* __restrict__ X, const double* __restrict__ Y,
                                         const double* Cin, double sign, double* Cout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int br = (wave >> 1) * 16, bc = (wave & 1) * 16;
  const int lr = lane & 15, lk = lane >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < kTile; kk += 4) {
    const double a = X[(br + lr) * kTileLd + kk + lk];
    const double b = Y[(kk + lk) * kTileLd + bc + lr];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = br + lk + 4 * i, c = bc + lr;
    const double base = Cin ? Cin[r * kTileLd + c] : 0.0;
    Cout[r * kTileLd + c] = base + sign * acc[i];
  }
}

// In-place inverse of a 32x32 SPD tile in LDS (Gauss-Jordan without pivoting).  The 32 pivot steps
// are a serial chain that sits on the critical path of every block step, so ONE wave keeps the
// whole tile in registers (lane l: column l & 31, rows 16 (l >> 5) .. +15) and runs the chain with
// wave shuffles only — no LDS round trips, no workgroup barriers inside the chain.
// v of lane `src` (wave-uniform, compile-time after unrolling) in every lane: two v_readlane_b32,
// no LDS-permute latency on the serial chain.
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void tile_inverse_lds(double* __restrict__ S, double* __restrict__ /*srow*/,
                                                 double* __restrict__ /*scol*/) {
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int c = lane & 31, h = lane >> 5;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = S[(h * 16 + i) * kTileLd + c];
#pragma unroll
    for (int p = 0; p < kTile; ++p) {
      const int ph = p >> 4, pi = p & 15;
      const double rowv = __shfl(a[pi], c + 32 * ph, 64);  // S[p][c]
      const double piv = readlane_f64(a[pi], p + 32 * ph);  // S[p][p]
      double ipiv = __builtin_amdgcn_rcp(piv);              // v_rcp_f64 + two Newton steps (full precision)
      ipiv = ipiv * (2.0 - piv * ipiv);
      ipiv = ipiv * (2.0 - piv * ipiv);
      const bool cp = c == p;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double c0 = readlane_f64(a[i], p), c1 = readlane_f64(a[i], p + 32);
        const double colv = h ? c1 : c0;  // S[r][p], r = 16 h + i
        const bool rp = (i == pi) && (h == ph);
        double v = a[i] - colv * rowv * ipiv;
        v = cp ? -colv * ipiv : v;
        v = rp ? rowv * ipiv : v;
        v = (rp && cp) ? ipiv : v;
        a[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) S[(h * 16 + i) * kTileLd + c] = a[i];
  }
  __syncthreads();
}

__device__ __forceinline__ void tile_load(const double* __restrict__ A, int ld, int bi, int bj, double* __restrict__ S) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = threadIdx.x + 256 * e;
    const int r = idx >> 5, c = idx & 31;
    S[r * kTileLd + c] = A[(size_t)(bi * kTile + r) * ld + bj * kTile + c];
  }
}
__device__ __forceinline__ void tile_store(double* __restrict__ A, int ld, int bi, int bj, const double* __restrict__ S) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = threadIdx.x + 256 * e;
    const int r = idx >> 5, c = idx & 31;
    A[(size_t)(bi * kTile + r) * ld + bj * kTile + c] = S[r * kTileLd + c];
  }
}

// transposed tile load: S = A(bi, bj)^T
__device__ __forceinline__ void tile_load_t(const double* __restrict__ A, int ld, int bi, int bj, double* __restrict__ S) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = threadIdx.x + 256 * e;
    const int r = idx >> 5, c = idx & 31;
    S[c * kTileLd + r] = A[(size_t)(bi * kTile + r) * ld + bj * kTile + c];
  }
}

// ---- symmetric block sweep (batched) ------------------------------------------------------------------------------
// The matrices are SPD, so the block Gauss-Jordan iteration is run in its SYMMETRIC form (the sweep operator):
//     B_kk = -P,   B_ik = A_ik P,   B_kj = P A_kj,   B_ij = A_ij - A_ik P A_kj          (P = A_kk^-1; i, j != k)
// keeps every iterate symmetric, so only the tiles of the lower triangle (bi >= bj) are computed and stored — half the
// tile products and half the traffic of the general iteration; a tile of the upper triangle is read as the transpose of
// its mirror image.  After all T sweeps the buffer holds  -A^-1  (gj_inv_at() below undoes the sign).
// One launch serves nz matrices (blockIdx.z; matrix z at in + z * zstride, T = Tz[z] tile rows or `Tu` when Tz is null);
// blockIdx.x enumerates the lower triangle of the LARGEST matrix.
static __global__ void __launch_bounds__(kBlock)
    k_gj_pivot0(const double* __restrict__ A, int ld, size_t zstride, double* __restrict__ pinv) {
  __shared__ double S[kTile * kTileLd], srow[kTile], scol[kTile];
  tile_load(A + zstride * blockIdx.z, ld, 0, 0, S);
  tile_inverse_lds(S, srow, scol);
  double* pz = pinv + (size_t)blockIdx.z * 2 * kTile * kTile;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = threadIdx.x + 256 * e;
    pz[idx] = S[(idx >> 5) * kTileLd + (idx & 31)];
  }
}

// pinv: [nz][2][32 x 32], the inverse of the pivot tile of step k in slot k & 1 (k_gj_pivot0 fills slot 0; the workgroup
// that produces tile (k+1, k+1) inverts it for the next step).  Matrices with fewer tile rows than the launch has steps
// are carried along unchanged (k >= T), so that all of them end in the same ping-pong buffer.
static __global__ void __launch_bounds__(kBlock)
    k_gj_sweep_step(const double* __restrict__ in, double* __restrict__ out, int ld, size_t zstride, const int* __restrict__ Tz,
                    int Tu, int k, double* __restrict__ pinv) {
  __shared__ double sP[kTile * kTileLd], sX[kTile * kTileLd], sY[kTile * kTileLd], sC[kTile * kTileLd],
      sM[kTile * kTileLd], srow[kTile], scol[kTile];
  const int z = blockIdx.z, T = Tz ? Tz[z] : Tu;
  // lower-triangle enumeration: t = bi (bi + 1) / 2 + bj
  const int t = blockIdx.x;
  int bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  while (bi * (bi + 1) / 2 > t) --bi;
  const int bj = t - bi * (bi + 1) / 2;
  if (bi >= T) return;
  const double* inz = in + zstride * z;
  double* outz = out + zstride * z;
  if (k >= T) {  // finished: carry the result along
    tile_load(inz, ld, bi, bj, sC);
    __syncthreads();
    tile_store(outz, ld, bi, bj, sC);
    return;
  }
  const double* pinv_in = pinv + ((size_t)z * 2 + (k & 1)) * kTile * kTile;
  double* pinv_out = pinv + ((size_t)z * 2 + ((k + 1) & 1)) * kTile * kTile;
  const double sgn = (bi == k && bj == k) ? -1.0 : 1.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = threadIdx.x + 256 * e;
    sP[(idx >> 5) * kTileLd + (idx & 31)] = sgn * pinv_in[idx];
  }
  double* res = sC;
  if (bi == k && bj == k) {
    res = sP;  // -P
    __syncthreads();
  } else if (bi == k) {  // bj < k: row tile, P A_kj
    tile_load(inz, ld, k, bj, sY);
    __syncthreads();
    tile_mma(sP, sY, nullptr, 1.0, sC);
  } else if (bj == k) {  // bi > k: column tile, A_ik P
    tile_load(inz, ld, bi, k, sX);
    __syncthreads();
    tile_mma(sX, sP, nullptr, 1.0, sC);
  } else {
    if (bi > k) tile_load(inz, ld, bi, k, sX); else tile_load_t(inz, ld, k, bi, sX);  // A_ik
    if (k > bj) tile_load(inz, ld, k, bj, sY); else tile_load_t(inz, ld, bj, k, sY);  // A_kj
    tile_load(inz, ld, bi, bj, sC);
    __syncthreads();
    tile_mma(sX, sP, nullptr, 1.0, sM);  // A_ik P
    __syncthreads();
    tile_mma(sM, sY, sC, -1.0, sC);  // A_ij - (A_ik P) A_kj   (each element read and written by its own lane)
  }
  __syncthreads();
  tile_store(outz, ld, bi, bj, res);
  if (bi == k + 1 && bj == k + 1) {  // next pivot
    tile_inverse_lds(res, srow, scol);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = threadIdx.x + 256 * e;
      pinv_out[idx] = res[(idx >> 5) * kTileLd + (idx & 31)];
    }
  }
}
inline int gj_tiles(int T) { return T * (T + 1) / 2; }

// Element (r, c) of A^-1 from a swept buffer: lower tile triangle stored, diagonal tiles complete, sign flipped.
__device__ __forceinline__ double gj_inv_at(const double* __restrict__ A, int ld, int r, int c) {
  const int tr = r / kTile, tc = c / kTile;
  if (tr == tc) return -0.5 * (A[(size_t)r * ld + c] + A[(size_t)c * ld + r]);
  return tr > tc ? -A[(size_t)r * ld + c] : -A[(size_t)c * ld + r];
}
// In place: the complete symmetric A^-1 from a swept buffer (dense direct path: the apply streams full rows).
static __global__ void __launch_bounds__(kBlock) k_gj_finish_full(double* __restrict__ A, int ld, int n) {
  const size_t nn = (size_t)n * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i % n);
    if (r < c) continue;  // one thread per unordered pair
    const double v = gj_inv_at(A, ld, r, c);
    A[(size_t)r * ld + c] = v;
    A[(size_t)c * ld + r] = v;
  }
}

// Dense assembly of (L_w + gauge) from the CSR-by-node incidence list: A[n][nbr] -= w (atomics: a
// pair of nodes may be linked by several edges), diagonal = lap_diag, identity on the padding.
static __global__ void __launch_bounds__(kBlock)
    k_dense_fill_offdiag(long nnz, const int* __restrict__ inc_row, const int* __restrict__ nbr,
                         const double* __restrict__ inc_w, int ld, double* __restrict__ A) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x)
    unsafeAtomicAdd(A + (size_t)inc_row[k] * ld + nbr[k], -inc_w[k]);
}
static __global__ void __launch_bounds__(kBlock)
    k_dense_fill_diag(int N, int Np, const double* __restrict__ lap_diag, int ld, double* __restrict__ A) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < Np; n += gridDim.x * blockDim.x)
    A[(size_t)n * ld + n] = n < N ? A[(size_t)n * ld + n] + lap_diag[n] : 1.0;  // += keeps self-loop terms
}

// y[n][0..3) (+)= sum_m Ainv[n][m] v[m][0..3): one wave per row.
static __global__ void __launch_bounds__(kBlock)
    k_dense_apply3(int N, int ld, const double* __restrict__ Ainv, const double* __restrict__ v,
                   double* __restrict__ y, int accumulate, const int* __restrict__ stop) {
  if (stop != nullptr && *stop) return;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  for (int n = wave; n < N; n += nwaves) {
    const double* row = Ainv + (size_t)n * ld;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int m = lane; m < N; m += 64) {
      const double w = row[m];
      a0 += w * v[3 * m];
      a1 += w * v[3 * m + 1];
      a2 += w * v[3 * m + 2];
    }
    a0 = group_sum<64>(a0);
    a1 = group_sum<64>(a1);
    a2 = group_sum<64>(a2);
    if (lane == 0) {
      if (accumulate) {
        y[3 * n] += a0;
        y[3 * n + 1] += a1;
        y[3 * n + 2] += a2;
      } else {
        y[3 * n] = a0;
        y[3 * n + 1] = a1;
        y[3 * n + 2] = a2;
      }
    }
  }
}

// r = b - Ax
static __global__ void __launch_bounds__(kBlock)
    k_dense_residual(long n, const double* __restrict__ b, const double* __restrict__ Ax, double* __restrict__ r) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) r[i] = b[i] - Ax[i];
}


// ---- one SPD system solved densely: the reduced camera systems of small GP / BA problems (gp.hip, ba_impl.hpp) ------------------
// y = sc (.) A (sc (.) v) (A: n x n, leading dimension ld; sc: optional diagonal scaling), one wave per row; with `b`: y = b - A v
static __global__ void __launch_bounds__(kBlock)
    k_dense_matvec(int n, int ld, const double* __restrict__ A, const double* __restrict__ v, const double* __restrict__ b,
                   const double* __restrict__ sc, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  for (int r = wave; r < n; r += nwaves) {
    const double* row = A + (size_t)r * ld;
    double acc = 0.0;
    if (sc != nullptr) {
      for (int m = lane; m < n; m += 64) acc += row[m] * (sc[m] * v[m]);
    } else {
      for (int m = lane; m < n; m += 64) acc += row[m] * v[m];
    }
    acc = wave_sum(acc);
    if (sc != nullptr) acc *= sc[r];
    if (lane == 0) y[r] = b != nullptr ? b[r] - acc : acc;
  }
}
static __global__ void __launch_bounds__(kBlock) k_dense_axpy(int n, const double* __restrict__ dx, double* __restrict__ x) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] += dx[i];
}
// sc_i = 1 / sqrt(S_ii) (1 on the padding and where the diagonal is not positive)
static __global__ void __launch_bounds__(kBlock) k_dense_diag_scale(int n, int ld, const double* __restrict__ S, double* __restrict__ sc) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ld; i += gridDim.x * blockDim.x) {
    const double d = i < n ? S[(size_t)i * ld + i] : 1.0;
    sc[i] = d > 0.0 ? 1.0 / sqrt(d) : 1.0;
  }
}
// out_ij = sc_i S_ij sc_j
static __global__ void __launch_bounds__(kBlock)
    k_dense_scale_copy(int ld, const double* __restrict__ S, const double* __restrict__ sc, double* __restrict__ out) {
  const size_t nn = (size_t)ld * ld;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += (size_t)gridDim.x * blockDim.x)
    out[i] = sc[i / ld] * S[i] * sc[i % ld];
}
// out[0] = |r|^2, out[1] = |b|^2 (one workgroup; n is a few thousand)
static __global__ void __launch_bounds__(kBlock)
    k_dense_norms(int n, const double* __restrict__ r, const double* __restrict__ b, double* __restrict__ out) {
  __shared__ double sm[2 * (kBlock / 64)];
  double a0 = 0.0, a1 = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    a0 += r[i] * r[i];
    a1 += b[i] * b[i];
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if ((threadIdx.x & 63) == 0) {
    sm[threadIdx.x >> 6] = a0;
    sm[kBlock / 64 + (threadIdx.x >> 6)] = a1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) t0 += sm[w], t1 += sm[kBlock / 64 + w];
    out[0] = t0;
    out[1] = t1;
  }
}
// ---- blocked Cholesky: no inverse of anything but a triangle is ever formed ---------------------------------------------------
// The sweep above ends with an explicit inverse, whose error grows like the SQUARE of the condition number — enough for the
// Laplacians and coarse matrices it was written for, not for a bundle adjustment at a large trust-region radius.  Measured on the
// mapper's second BA of a 300-image ring, whose equilibrated reduced systems have condition numbers 2.8e4 ... 1.3e6 (numpy on the
// dumped matrices, tools/exp_capture_ba_verbose.py: SPD, symmetric to 3e-16, LU residual 1e-13): the sweep's first residual is
// 1.6e-7 / 4e-3 / 9.7 at radius 1e4 / 3e4 / 9e4; a block Gaussian elimination that still multiplied by explicitly inverted 32 x 32
// pivots reached 5.5e-10 / 4.3e-5 / 7e-4 and stalled two radius steps later (the pivots are Schur complements as ill-conditioned
// as the matrix, and their inverses enter every trailing update).  The factorisation the reference's SPARSE_SCHUR performs is
// backward stable because it only ever divides by a TRIANGLE:  A = L L^T by 32-column panels,
//     L_kk = chol(A_kk), W_k = L_kk^-1 (one workgroup, in LDS),   L_ik = A_ik W_k^T (i > k),   A_ij -= L_ik L_jk^T (i >= j > k)
// two launches per step, the matrix cores in both; the workgroup that updates tile (k+1, k+1) factorises it for the next step.
// The solve is one workgroup walking the tiles (n^2 multiply-adds: microseconds at these sizes):
//     forward   y_k = W_k (b_k - sum_{j < k} L_kj y_j),     backward   x_k = W_k^T (y_k - sum_{i > k} L_ik^T x_i).
// S: SPD tile in LDS (leading dimension kTileLd), overwritten by its Cholesky factor (lower triangle); W: its inverse (lower
// triangle, zeros above).  All 256 threads of the workgroup.
__device__ __forceinline__ void tile_cholesky_lds(double* __restrict__ S, double* __restrict__ W) {
  const int tid = threadIdx.x;
  __syncthreads();
  for (int p = 0; p < kTile; ++p) {
    if (tid == 0) S[p * kTileLd + p] = sqrt(S[p * kTileLd + p]);
    __syncthreads();
    if (tid > p && tid < kTile) S[tid * kTileLd + p] /= S[p * kTileLd + p];
    __syncthreads();
    for (int e = tid; e < kTile * kTile; e += kBlock) {
      const int r = e >> 5, c = e & 31;
      if (c > p && r >= c) S[r * kTileLd + c] -= S[r * kTileLd + p] * S[c * kTileLd + p];
    }
    __syncthreads();
  }
  for (int e = tid; e < kTile * kTile; e += kBlock) W[(e >> 5) * kTileLd + (e & 31)] = 0.0;
  __syncthreads();
  if (tid < kTile) {  // column `tid` of L^-1 by forward substitution (each thread reads only what it wrote)
    const int c = tid;
    W[c * kTileLd + c] = 1.0 / S[c * kTileLd + c];
    for (int r = c + 1; r < kTile; ++r) {
      double acc = 0.0;
      for (int q = c; q < r; ++q) acc += S[r * kTileLd + q] * W[q * kTileLd + c];
      W[r * kTileLd + c] = -acc / S[r * kTileLd + r];
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void tile_store_w(double* __restrict__ Wall, int k, const double* __restrict__ W) {
  double* w = Wall + (size_t)k * kTile * kTile;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = threadIdx.x + 256 * e;
    w[idx] = W[(idx >> 5) * kTileLd + (idx & 31)];
  }
}
static __global__ void __launch_bounds__(kBlock) k_ch_pivot0(const double* __restrict__ A, int ld, double* __restrict__ Wall) {
  __shared__ double sC[kTile * kTileLd], sW[kTile * kTileLd];
  tile_load(A, ld, 0, 0, sC);
  tile_cholesky_lds(sC, sW);
  tile_store_w(Wall, 0, sW);
}
static __global__ void __launch_bounds__(kBlock)
    k_ch_panel(const double* __restrict__ A, double* __restrict__ L, int ld, int k, const double* __restrict__ Wall) {
  __shared__ double sY[kTile * kTileLd], sX[kTile * kTileLd], sC[kTile * kTileLd];
  const int i = k + 1 + blockIdx.x;
  const double* w = Wall + (size_t)k * kTile * kTile;
#pragma unroll
  for (int e = 0; e < 4; ++e) {  // sY = W_k^T
    const int idx = threadIdx.x + 256 * e;
    sY[(idx & 31) * kTileLd + (idx >> 5)] = w[idx];
  }
  tile_load(A, ld, i, k, sX);
  __syncthreads();
  tile_mma(sX, sY, nullptr, 1.0, sC);  // A_ik L_kk^-T
  __syncthreads();
  tile_store(L, ld, i, k, sC);
}
static __global__ void __launch_bounds__(kBlock)
    k_ch_update(double* __restrict__ A, const double* __restrict__ L, int ld, int k, double* __restrict__ Wall) {
  __shared__ double sX[kTile * kTileLd], sY[kTile * kTileLd], sC[kTile * kTileLd];
  // lower-triangle enumeration of the trailing matrix: t = bi (bi + 1) / 2 + bj
  const int t = blockIdx.x;
  int bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  while (bi * (bi + 1) / 2 > t) --bi;
  const int bj = t - bi * (bi + 1) / 2;
  const int i = k + 1 + bi, j = k + 1 + bj;
  tile_load(L, ld, i, k, sX);    // L_ik
  tile_load_t(L, ld, j, k, sY);  // L_jk^T
  tile_load(A, ld, i, j, sC);
  __syncthreads();
  tile_mma(sX, sY, sC, -1.0, sC);
  __syncthreads();
  tile_store(A, ld, i, j, sC);
  if (bi == 0 && bj == 0) {  // the next diagonal tile: factorise it (sX is free: its inverse goes there)
    tile_cholesky_lds(sC, sX);
    tile_store_w(Wall, k + 1, sX);
  }
}
// out (+)= sc (.) A^-1 (sc (.) b) from the factor (L: the tiles below the diagonal, Wall: the inverted diagonal factors); b and
// out have n entries, the work vector y has ld; one workgroup of 1024 threads.
static __global__ void __launch_bounds__(1024)
    k_ch_solve(int ld, int n, const double* __restrict__ L, const double* __restrict__ Wall, const double* __restrict__ sc,
               const double* __restrict__ b, double* __restrict__ y, double* __restrict__ out, int accumulate) {
  __shared__ double sk[kTile], part[32][kTile + 1];
  const int T = ld / kTile, tid = threadIdx.x;
  for (int r = tid; r < ld; r += blockDim.x) y[r] = r < n ? sc[r] * b[r] : 0.0;
  __syncthreads();
  for (int k = 0; k < T; ++k) {  // forward
    if (tid < kTile) {
      const double* w = Wall + (size_t)k * kTile * kTile + tid * kTile;
      double acc = 0.0;
      for (int q = 0; q <= tid; ++q) acc += w[q] * y[k * kTile + q];
      sk[tid] = acc;
    }
    __syncthreads();
    if (tid < kTile) y[k * kTile + tid] = sk[tid];
    for (int r = (k + 1) * kTile + tid; r < ld; r += blockDim.x) {
      const double* m = L + (size_t)r * ld + k * kTile;
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < kTile; ++c) acc += m[c] * sk[c];
      y[r] -= acc;
    }
    __syncthreads();
  }
  for (int k = T - 1; k >= 0; --k) {  // backward: column c of the tiles below the diagonal, 32 row groups
    const int c = tid & 31, g = tid >> 5;
    double acc = 0.0;
    for (int r = (k + 1) * kTile + g; r < ld; r += 32) acc += L[(size_t)r * ld + k * kTile + c] * y[r];
    part[g][c] = acc;
    __syncthreads();
    if (tid < kTile) {
      double s0 = y[k * kTile + tid];
      for (int q = 0; q < 32; ++q) s0 -= part[q][tid];
      sk[tid] = s0;
    }
    __syncthreads();
    if (tid < kTile) {  // W_k^T s
      const double* w = Wall + (size_t)k * kTile * kTile;
      double x = 0.0;
      for (int q = tid; q < kTile; ++q) x += w[q * kTile + tid] * sk[q];
      y[k * kTile + tid] = x;
    }
    __syncthreads();
  }
  for (int r = tid; r < n; r += blockDim.x) out[r] = (accumulate ? out[r] : 0.0) + sc[r] * y[r];
}

// x = S0^-1 rhs: S0 (ld x ld, SPD, identity on the padding rows / columns beyond n; left intact) is equilibrated symmetrically
// (unit diagonal — Ceres' Jacobi scaling: the unknowns of a bundle adjustment differ by many orders of magnitude), factorised
// (above) and solved; then iterative refinement x += S0^-1 (rhs - S0 x) against the matrix itself until
// |rhs - S0 x| <= tol |rhs|.  The residual is CHECKED (one small read-back per step): returns false when it has not reached tol
// after max_refine steps or stops halving — the caller then solves this system by PCG, so a failure here costs time, never an
// answer.  bufA / bufB: ld x ld scratch; P: (ld / 32) x 32 x 32; r, dx, sc: ld each; nrm: 2 doubles on the device.
inline bool dense_spd_solve(hipStream_t s, int n, int ld, const double* S0, double* bufA, double* bufB, double* P,
                            double* r, double* dx, double* sc, double* nrm, const double* rhs, double* x, double tol, int max_refine = 6) {
  const int T = ld / kTile;
  const size_t nn = (size_t)ld * ld;
  hipLaunchKernelGGL(k_dense_diag_scale, dim3(grid_for((size_t)ld, kBlock)), dim3(kBlock), 0, s, n, ld, S0, sc);
  hipLaunchKernelGGL(k_dense_scale_copy, dim3(grid_wide(nn, kBlock, 1 << 12)), dim3(kBlock), 0, s, ld, S0, (const double*)sc, bufA);
  hipLaunchKernelGGL(k_ch_pivot0, dim3(1), dim3(kBlock), 0, s, (const double*)bufA, ld, P);
  for (int k = 0; k + 1 < T; ++k) {
    const int m = T - k - 1;
    hipLaunchKernelGGL(k_ch_panel, dim3(m), dim3(kBlock), 0, s, (const double*)bufA, bufB, ld, k, (const double*)P);
    hipLaunchKernelGGL(k_ch_update, dim3(m * (m + 1) / 2), dim3(kBlock), 0, s, bufA, (const double*)bufB, ld, k, P);
  }
  const int gridR = grid_wide((size_t)n, kBlock / 64, 1 << 12);
  hipLaunchKernelGGL(k_ch_solve, dim3(1), dim3(1024), 0, s, ld, n, (const double*)bufB, (const double*)P, (const double*)sc, rhs, dx, x, 0);
  const bool verbose = std::getenv("GSFM_VERBOSE") != nullptr;
  double prev = -1.0;
  for (int it = 0; it <= max_refine; ++it) {
    hipLaunchKernelGGL(k_dense_matvec, dim3(gridR), dim3(kBlock), 0, s, n, ld, S0, (const double*)x, rhs, (const double*)nullptr, r);
    hipLaunchKernelGGL(k_dense_norms, dim3(1), dim3(kBlock), 0, s, n, (const double*)r, rhs, nrm);
    double h[2];
    GSFM_HIP_CHECK(hipMemcpyAsync(h, nrm, sizeof h, hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    const double rel = h[1] > 0.0 ? std::sqrt(h[0] / h[1]) : 0.0;
    if (verbose) fprintf(stderr, "[gsfm dense] n %d, %d refinement steps: |rhs - S x| / |rhs| = %.3e\n", n, it, rel);
    if (!std::isfinite(rel)) return false;
    if (rel <= tol) return true;
    if (it == max_refine || (prev >= 0.0 && rel > 0.5 * prev)) return false;  // not converging (fast enough)
    prev = rel;
    hipLaunchKernelGGL(k_ch_solve, dim3(1), dim3(1024), 0, s, ld, n, (const double*)bufB, (const double*)P, (const double*)sc, (const double*)r,
                       dx, x, 1);
  }
  return false;
}

// The same with the EXPLICIT inverse of the symmetric block sweep (one launch per step instead of two, the solves two dense
// matrix-vector products instead of a single-workgroup substitution) — faster where the condition number allows it: global
// positioning's reduced systems (no rotations, no intrinsics).  pinv: [2][32 x 32].
inline bool dense_spd_solve_by_inverse(hipStream_t s, int n, int ld, const double* S0, double* bufA, double* bufB, double* pinv /* [2][32 x 32] */,
                            double* r, double* dx, double* sc, double* nrm, const double* rhs, double* x, double tol, int max_refine = 8) {
  const int T = ld / kTile;
  const size_t nn = (size_t)ld * ld;
  double *cur = bufA, *oth = bufB;
  hipLaunchKernelGGL(k_dense_diag_scale, dim3(grid_for((size_t)ld, kBlock)), dim3(kBlock), 0, s, n, ld, S0, sc);
  hipLaunchKernelGGL(k_dense_scale_copy, dim3(grid_wide(nn, kBlock, 1 << 12)), dim3(kBlock), 0, s, ld, S0, (const double*)sc, cur);
  hipLaunchKernelGGL(k_gj_pivot0, dim3(1), dim3(kBlock), 0, s, cur, ld, (size_t)0, pinv);
  for (int k = 0; k < T; ++k) {
    hipLaunchKernelGGL(k_gj_sweep_step, dim3(gj_tiles(T)), dim3(kBlock), 0, s, cur, oth, ld, (size_t)0, (const int*)nullptr, T, k, pinv);
    std::swap(cur, oth);
  }
  hipLaunchKernelGGL(k_gj_finish_full, dim3(grid_wide(nn, kBlock, 1 << 12)), dim3(kBlock), 0, s, cur, ld, ld);
  const int gridR = grid_wide((size_t)n, kBlock / 64, 1 << 12), gridV = grid_for((size_t)n, kBlock);
  hipLaunchKernelGGL(k_dense_matvec, dim3(gridR), dim3(kBlock), 0, s, n, ld, (const double*)cur, rhs, (const double*)nullptr, (const double*)sc, x);
  const bool verbose = std::getenv("GSFM_VERBOSE") != nullptr;
  double prev = -1.0;
  for (int it = 0; it <= max_refine; ++it) {
    hipLaunchKernelGGL(k_dense_matvec, dim3(gridR), dim3(kBlock), 0, s, n, ld, S0, (const double*)x, rhs, (const double*)nullptr, r);
    hipLaunchKernelGGL(k_dense_norms, dim3(1), dim3(kBlock), 0, s, n, (const double*)r, rhs, nrm);
    double h[2];
    GSFM_HIP_CHECK(hipMemcpyAsync(h, nrm, sizeof h, hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    const double rel = h[1] > 0.0 ? std::sqrt(h[0] / h[1]) : 0.0;
    if (verbose) fprintf(stderr, "[gsfm dense] n %d, %d refinement steps: |rhs - S x| / |rhs| = %.3e\n", n, it, rel);
    if (!std::isfinite(rel)) return false;
    if (rel <= tol) return true;
    if (it == max_refine || (prev >= 0.0 && rel > 0.5 * prev)) return false;  // not converging (fast enough): the inverse is no good
    prev = rel;
    hipLaunchKernelGGL(k_dense_matvec, dim3(gridR), dim3(kBlock), 0, s, n, ld, (const double*)cur, (const double*)r, (const double*)nullptr,
                       (const double*)sc, dx);
    hipLaunchKernelGGL(k_dense_axpy, dim3(gridV), dim3(kBlock), 0, s, n, (const double*)dx, x);
  }
  return false;
}

// ---- PCG preconditioned by a STALE dense inverse --------------------------------------------------
// The IRLS systems (L_w + gauge) x = rhs differ from the L1-stage matrix only by the edge weights, and
// the Geman-McClure weights of the inliers are nearly equal: (L_1 + gauge)^-1 is an excellent
// preconditioner (C2: 8 iterations to 1e-10), so an IRLS iteration costs a few (dense apply + SpMV +
// update) launches instead of a 1.2 ms re-inversion.  Single-reduction (Chronopoulos-Gear) recurrences;
// the vectors are 3N <= 6144 doubles, so ONE workgroup owns every scalar and the convergence flag.
struct DpcgState {
  int done;   // first member: k_dense_apply3's `stop` pointer aliases it
  int iters;
  int bad;
  int pad;
  double gamma_old, alpha_old, bb, rr;
};

static __global__ void __launch_bounds__(1024)
    k_dpcg_init(int n3, const double* __restrict__ b, double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                double* __restrict__ s, DpcgState* st) {
  __shared__ double smem[16];
  double acc[1] = {0.0};
  for (int i = threadIdx.x; i < n3; i += blockDim.x) {
    const double bi = b[i];
    x[i] = 0.0;
    r[i] = bi;
    p[i] = 0.0;
    s[i] = 0.0;
    acc[0] += bi * bi;
  }
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) {
    st->done = acc[0] > 0.0 ? 0 : 1;  // b = 0 -> x = 0
    st->iters = 0;
    st->bad = 0;
    st->gamma_old = 0.0;
    st->alpha_old = 0.0;
    st->bb = acc[0];
    st->rr = acc[0];
  }
}

// The same start on many workgroups (block-preconditioned PCG, 3N = 30 000 .. 100 000 doubles: one workgroup takes 20 us):
// vectors and per-block |b|^2 partials here, the state in k_dpcg_init_fin.
static __global__ void __launch_bounds__(kBlock)
    k_dpcg_init_mb(int n3, const double* __restrict__ b, double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                   double* __restrict__ s, double* __restrict__ part) {
  __shared__ double smem[4];
  double acc[1] = {0.0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x) {
    const double bi = b[i];
    x[i] = 0.0;
    r[i] = bi;
    p[i] = 0.0;
    s[i] = 0.0;
    acc[0] += bi * bi;
  }
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}
static __global__ void __launch_bounds__(64) k_dpcg_init_fin(const double* __restrict__ part, int nparts, DpcgState* st) {
  double v = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) v += part[i];
  v = wave_sum(v);
  if (threadIdx.x == 0) {
    st->done = v > 0.0 ? 0 : 1;  // b = 0 -> x = 0
    st->iters = 0;
    st->bad = 0;
    st->gamma_old = 0.0;
    st->alpha_old = 0.0;
    st->bb = v;
    st->rr = v;
  }
}

// u = M r and w = A u are in; gamma = r.u, delta = w.u, then the vector recurrences and |r|^2.
static __global__ void __launch_bounds__(1024)
    k_dpcg_update(int n3, double tol2, const double* __restrict__ u, const double* __restrict__ w, double* __restrict__ x,
                  double* __restrict__ r, double* __restrict__ p, double* __restrict__ s, DpcgState* st) {
  __shared__ double smem[16 * 2 + 2];
  if (st->done) return;
  double acc[2] = {0.0, 0.0};
  for (int i = threadIdx.x; i < n3; i += blockDim.x) {
    acc[0] += r[i] * u[i];
    acc[1] += w[i] * u[i];
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    smem[32] = acc[0];
    smem[33] = acc[1];
  }
  __syncthreads();
  const double gamma = smem[32], delta = smem[33];
  const bool first = st->iters == 0;
  const double beta = first ? 0.0 : gamma / st->gamma_old;
  const double denom = first ? delta : delta - beta * gamma / st->alpha_old;
  const double alpha = gamma / denom;
  const bool ok = denom > 0.0 && gamma > 0.0 && isfinite(alpha);
  __syncthreads();
  double rr[1] = {0.0};
  if (ok) {
    for (int i = threadIdx.x; i < n3; i += blockDim.x) {
      const double pi = u[i] + beta * p[i];
      const double si = w[i] + beta * s[i];
      p[i] = pi;
      s[i] = si;
      x[i] += alpha * pi;
      const double ri = r[i] - alpha * si;
      r[i] = ri;
      rr[0] += ri * ri;
    }
  }
  block_sum<1>(rr, smem);
  if (threadIdx.x == 0) {
    if (!ok) {
      st->bad = 1;
      st->done = 1;
    } else {
      st->gamma_old = gamma;
      st->alpha_old = alpha;
      st->iters += 1;
      st->rr = rr[0];
      if (rr[0] <= tol2 * st->bb) st->done = 1;
    }
  }
}


// ---- block-diagonal dense preconditioner for 2048 < N <= 32768 (ra.hip: bd_*) -----------------------------
// Nodes are relabelled in BFS order at setup, so index-contiguous blocks of nb <= 2048 nodes capture almost every
// edge of a view graph with any locality; each diagonal block of (L_w + gauge) is inverted with the same tiled
// Gauss-Jordan sweep, and M = blockdiag(A_bb^-1) preconditions the PCG (C4 ring graph: 38 iterations instead of
// 279 with Jacobi).
static __global__ void __launch_bounds__(kBlock)
    k_bd_fill_offdiag(long nnz, const int* __restrict__ inc_row, const int* __restrict__ nbr, const double* __restrict__ inc_w,
                      int b0, int nb, double* __restrict__ A) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x) {
    const int r = inc_row[k] - b0, c = nbr[k] - b0;
    if ((unsigned)r < (unsigned)nb && (unsigned)c < (unsigned)nb) unsafeAtomicAdd(A + (size_t)r * nb + c, -inc_w[k]);
  }
}
static __global__ void __launch_bounds__(kBlock)
    k_bd_fill_diag(int N, int b0, int nb, const double* __restrict__ lap_diag, double* __restrict__ A) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
    A[(size_t)i * nb + i] = (b0 + i) < N ? A[(size_t)i * nb + i] + lap_diag[b0 + i] : 1.0;
}

// Multi-block single-reduction PCG for the block-diagonal preconditioner (3 launches per iteration, scalars and the
// convergence decision on the device, partial sums in fixed per-block slots re-reduced in a fixed order):
//   k_bd_apply3 : [test |r|^2 of the previous update]  u = blockdiag(inv_b) r
//   k_bd_spmv   : w = A u, partials of gamma = r.u and delta = w.u
//   k_bd_update : alpha / beta from the partials, p = u + beta p, s = w + beta s, x += alpha p, r -= alpha s, |r|^2 partials
constexpr int kBdUpdateBlocks = 64;
struct BdScal {
  double gamma, alpha;
};

__device__ __forceinline__ double bd_reduce1(const double* __restrict__ part, int n, double* smem /* >= 5 */) {
  double acc[1] = {0.0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc[0] += part[i];
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) smem[4] = acc[0];
  __syncthreads();
  const double v = smem[4];
  __syncthreads();
  return v;
}

// v / y are [N][3] in the (BFS) node order the blocks are cut in; one wave per row.
// The block inverses are only a preconditioner, so they are stored in fp32 (symmetrised before rounding: CG needs a
// symmetric M) — the apply is a pure stream of the blocks, half the bytes is nearly half the time; sums stay in f64.
static __global__ void __launch_bounds__(kBlock)
    k_bd_to_f32(int nb, const double* __restrict__ swept, size_t zstride, float* __restrict__ out) {
  const size_t nn = (size_t)nb * nb;
  const double* A = swept + zstride * blockIdx.y;
  float* o = out + nn * blockIdx.y;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += (size_t)gridDim.x * blockDim.x)
    o[i] = (float)gj_inv_at(A, nb, (int)(i / nb), (int)(i % nb));
}

static __global__ void __launch_bounds__(kBlock)
    k_bd_apply3(int N, int nb, const float* __restrict__ inv, const double* __restrict__ v, double* __restrict__ y, int it,
                double tol2, const double* __restrict__ rpart, DpcgState* st) {
  __shared__ double smem[5];
  if (st->done) return;
  if (it > 0) {
    const double rr = bd_reduce1(rpart, kBdUpdateBlocks, smem);
    if (rr <= tol2 * st->bb) {  // every block takes the same decision from the same slots
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rr = rr;
        st->done = 1;
      }
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  for (int n = wave; n < N; n += nwaves) {
    const int b = n / nb, b0 = b * nb;
    const int cols = min(nb, N - b0);
    const float* row = inv + (size_t)b * nb * nb + (size_t)(n - b0) * nb;
    const double* vb = v + 3 * (size_t)b0;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int m = lane; m < cols; m += 64) {
      const double w = (double)row[m];
      a0 += w * vb[3 * m];
      a1 += w * vb[3 * m + 1];
      a2 += w * vb[3 * m + 2];
    }
    a0 = group_sum<64>(a0);
    a1 = group_sum<64>(a1);
    a2 = group_sum<64>(a2);
    if (lane == 0) {
      y[3 * (size_t)n] = a0;
      y[3 * (size_t)n + 1] = a1;
      y[3 * (size_t)n + 2] = a2;
    }
  }
}

static __global__ void __launch_bounds__(kBlock)
    k_bd_update(int n3, int nslots, const double* __restrict__ dpart, const double* __restrict__ u, const double* __restrict__ w,
                double* __restrict__ x, double* __restrict__ r, double* __restrict__ p, double* __restrict__ s, int it,
                const BdScal* __restrict__ scal_in, BdScal* __restrict__ scal_out, double* __restrict__ rpart, DpcgState* st) {
  __shared__ double smem[4 * 2 + 2];
  if (st->done) return;
  double gd[2];
  reduce_partials<2>(dpart, nslots, gd, smem);
  const double gamma = gd[0], delta = gd[1];
  const double beta = it == 0 ? 0.0 : gamma / scal_in->gamma;
  const double denom = it == 0 ? delta : delta - beta * gamma / scal_in->alpha;
  const double alpha = gamma / denom;
  const bool ok = denom > 0.0 && gamma > 0.0 && isfinite(alpha);
  double rr[1] = {0.0};
  if (ok) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x) {
      const double pi = u[i] + beta * p[i];
      const double si = w[i] + beta * s[i];
      p[i] = pi;
      s[i] = si;
      x[i] += alpha * pi;
      const double ri = r[i] - alpha * si;
      r[i] = ri;
      rr[0] += ri * ri;
    }
  }
  block_sum<1>(rr, smem);
  if (threadIdx.x == 0) {
    rpart[blockIdx.x] = rr[0];
    if (blockIdx.x == 0) {
      if (!ok) {
        st->bad = 1;
        st->done = 1;
      } else {
        scal_out->gamma = gamma;
        scal_out->alpha = alpha;
        st->iters = it + 1;
      }
    }
  }
}

// out[order[p]] = in[p]  (3 doubles per node): results back from BFS order to the caller's node order
static __global__ void __launch_bounds__(kBlock)
    k_unpermute3(int N, const int* __restrict__ order, const double* __restrict__ in, double* __restrict__ out) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x) {
    const size_t n = (size_t)order[p];
    out[3 * n] = in[3 * (size_t)p];
    out[3 * n + 1] = in[3 * (size_t)p + 1];
    out[3 * n + 2] = in[3 * (size_t)p + 2];
  }
}

}  // namespace gsfm
