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
// Convergence (|r|^2 <= tol^2 |b|^2) is tested by the vector kernel that produces an iterate (cg_status_tail: the
// workgroup that finishes last adds the |r|^2 partials); it raises status->done and every later kernel of the stream
// returns at once, so the host only polls the status every few iterations.
//
// Multi-rank: `apply` produces this rank's share of w and of delta; the solver all-reduces
// w[0..n] (delta travels in w[n]) over RCCL, then k_cg_update runs redundantly on every rank.
//
// Deflation of known near-null modes (CgDeflation below): the block-Jacobi-preconditioned reduced systems of BA / GP have
// a tight spectrum plus the similarity gauge of the scene — a handful of eigenvalues 1e-3 ... 1e-6 whose eigenvectors
// are known in closed form.  Deflated PCG (Saad, Yeung, Erhel, Guyomarc'h 2000) on span W:
//     E = W^T A W,   x = W E^-1 W^T b + x',   A x' = b2 := b - A W E^-1 W^T b,
//     every preconditioned residual is projected:  z_p = z - W y,  y = E^-1 (A W)^T z.
// The projection costs NO launch here: A is applied to the unprojected z, and k_cg_update, which already visits every
// element, uses  z_p = z - W y,  w_p = A z_p = w - (A W) y,  r.z_p = r.z - (W^T r).y,  z_p.w_p = z.w - y.((A W)^T z)
// (A symmetric, E y = (A W)^T z); the 2k dot products (A W)^T z and W^T r of the NEXT iterate are accumulated by the same
// kernel next to its r.z / r.r partials.  Same system, same stopping rule (relative to |b2| <= |b|).
#pragma once

#include "device.hpp"

namespace gsfm {

constexpr int kCgMaxBlocks = 512;  // partial-slot capacity of the vector kernels (k_cg_init / k_cg_update)
constexpr int kCgUpdateBlocks = 128;  // their grid: every block re-reduces the slots of all blocks, so few, fat blocks
constexpr int kCgMaxModes = 8;     // deflated modes per solve (CgDeflation)
constexpr int kMaxApplySlots = 4096;  // cap of the delta partial slots one apply kernel may write
constexpr int kCgMaxRecycle = 16;  // recycled Ritz vectors in the additive coarse space of a solve (CgRecycle)
constexpr int kCgHistCap = 128;    // Lanczos steps of a solve that are recorded for the next harvest

struct CgStatus {
  int done;
  int iters;
  int bad;  // non-finite or non-positive curvature encountered
  int pad;  // ticket of cg_status_tail (zero between kernels)
  double bb;  // |b|^2
  double rr;  // |r|^2 of the last tested iterate
  double gamma;  // single-workgroup mode: r.z of the current iterate
};
struct CgScal {
  double gamma;  // r.z of the previous iteration
  double alpha;  // step length of the previous iteration
};

// Device view.  Unknown layout: [PB per camera | IW per intrinsics block]; block-Jacobi blocks are
// the PB x PB camera blocks (minv + PB*PB*n) followed by IW x IW intrinsics blocks (IW = 8; 16 in the wide BA unit,
// ba_wide.hip — the joint and single-workgroup kernels below exist for IW = 8 / no intrinsics only).
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
  double* w = nullptr;         // [n + 2]; [n + 1 + kCgMaxRecycle] where a CgRecycle is passed to cg_solve
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
  // operator probe (forming A W): cg_converged() answers "not finished" and leaves the status alone
  int probe = 0;
  // deflation, device view (dk == 0: plain PCG); set by cg_solve from a CgDeflation
  int dk = 0;
  const double* dW = nullptr;      // [dk][n]
  const double* dAW = nullptr;     // [dk][n]
  const double* dsmall = nullptr;  // E^-1 [8][8] | y0 [8] | ok
  double* dcd = nullptr;           // [2][kCgMaxBlocks][2 * kCgMaxModes]: ((A W)^T z | W^T r) partials by iteration parity
  // recycled Ritz vectors, device view (rk == 0: none); set by cg_solve from a CgRecycle.  z = M^-1 r + sum_j u_j (u_j.r) / theta_j
  int rk = 0;                      // slots in use: 0 .. rk-1 (an empty slot has rcoef == 0)
  int rhist = 0;                   // record this solve's Lanczos history (zhist, hist)
  int r_nslots = 0;                // partial slots of u_j . w the apply kernels write per iteration
  const double* rU = nullptr;      // [kCgMaxRecycle][n]
  const double* rG = nullptr;      // [kCgMaxRecycle][kCgMaxModes]: u_j . (A W_i) of this solve's deflated modes
  double* rstate = nullptr;        // [2][2 kCgMaxRecycle]: (c_j = u_j . r | u_j . s) by iteration parity
  double* ruw = nullptr;           // [r_nslots][kCgMaxRecycle] partials of u_j . w, written by the last apply kernel
  double* zhist = nullptr;         // [kCgHistCap][n]: the projected z of iteration it
  double* hist = nullptr;          // [kCgHistCap][2]: (gamma, alpha) of iteration it
  double rcoef[kCgMaxRecycle] = {};  // 1 / theta_j
};

// ---- deflation, per-element pieces (all no-ops for v.dk == 0) ----------------------------------------------------
// One PCG step's scalars, identical in every thread of every block (all of them re-reduce the same partial slots in the
// same order).  y = E^-1 ((A W)^T z): the coefficients of the deflation projection (zero without deflation).
struct CgStep {
  double gamma, delta, alpha, beta;
  bool ok;
  double y[kCgMaxModes];
  double a[kCgMaxRecycle];  // coefficients of the recycled vectors in the NEW z: (u_j . r_new) / theta_j
};

constexpr int kCgStepSmemBase = 8 * 32 + 32 + 16 + kCgMaxModes * kCgMaxModes;
// + recycled vectors: [64][16] group partials of u_j . w | [16] a_j | [2 x 16] previous (c_j, u_j . s) | [16][8] G
constexpr int kCgStepSmem = kCgStepSmemBase + 64 * kCgMaxRecycle + kCgMaxRecycle + 2 * kCgMaxRecycle + kCgMaxRecycle * kCgMaxModes;

// Prologue of the vector update of iteration `it`: totals of all partial slots (r.z | r.r of the previous update, delta of
// this apply, the 2k deflation dot products), then alpha / beta.  Returns false when the solve is finished.
// The slot arrays are summed "transposed" — thread (g, l) adds column l of the slots b = g, g + 8, ... — so the totals
// need no cross-lane shuffles at all (first version: a 19-value block reduction, 230 LDS permutes per wave; the kernel
// took 50 us in deflated solves).  Only delta, one value spread over up to 4 096 slots, goes through a wave reduction.
// The loads are issued before the `done` test so that they overlap its round trip.
__device__ __forceinline__ bool cg_step_prologue(const CgVec& v, int it, CgStep& o, double* smem /* >= kCgStepSmem */) {
  constexpr int KD = 2 * kCgMaxModes;
  const int par = it & 1;
  const int g = threadIdx.x >> 5, l = threadIdx.x & 31;  // 8 groups of 32 lanes: column l of the slots b = g (mod 8)
  double col = 0.0, dl = 0.0;
  {
    const double* vp = v.vpart + (size_t)par * kCgMaxBlocks * 2;
    const double* cp = v.dcd + (size_t)par * kCgMaxBlocks * KD;
    if (l < 2) {
      for (int b = g; b < v.nb_update; b += 8) col += vp[2 * b + l];
    } else if (v.dk && l >= 3 && l < 3 + KD) {
      for (int b = g; b < v.nb_update; b += 8) col += cp[(size_t)b * KD + (l - 3)];
    }
    if (!v.delta_in_w)
      for (int b = threadIdx.x; b < v.nb_apply; b += blockDim.x) dl += v.dpart[b];
  }
  double* sg = smem;                 // [8][32] group partials
  double* stot = smem + 8 * 32;      // [32] totals (column 2 = delta)
  double* sb = smem + 8 * 32 + 32;   // [16] the step's scalars
  double* sE = sb + 16;              // E^-1, staged by the first 64 threads
  double* su = smem + kCgStepSmemBase;          // [64][kCgMaxRecycle] group partials of u_j . w
  double* sa = su + 64 * kCgMaxRecycle;         // [kCgMaxRecycle] a_j
  double* sc = sa + kCgMaxRecycle;              // [2 kCgMaxRecycle] (c_j | u_j . s) of the previous iteration
  double* sG = sc + 2 * kCgMaxRecycle;          // [kCgMaxRecycle][kCgMaxModes]
  typedef double cg_d4v __attribute__((ext_vector_type(4)));
  cg_d4v ucol = {0.0, 0.0, 0.0, 0.0};
  double uwd = 0.0;  // multi-rank: the all-reduced u_j . w arrived behind delta in w[n + 1 + j]
  if (v.rk && v.delta_in_w) {
    if (threadIdx.x < kCgMaxRecycle) uwd = v.w[v.n + 1 + threadIdx.x];
    if (threadIdx.x < 2 * kCgMaxRecycle) sc[threadIdx.x] = v.rstate[(size_t)par * 2 * kCgMaxRecycle + threadIdx.x];
    if (threadIdx.x < kCgMaxRecycle * kCgMaxModes) sG[threadIdx.x] = v.dk ? v.rG[threadIdx.x] : 0.0;
  } else if (v.rk) {
    // a slot row is 16 doubles: thread (group = threadIdx >> 2, quarter = threadIdx & 3) adds its 32-byte quarter of the rows
    // b = group (mod 64) — wide loads, all in flight together: the slots are one or two round trips, not one per row
    const int uq = threadIdx.x & 3, ug = threadIdx.x >> 2;
    const cg_d4v* rows = reinterpret_cast<const cg_d4v*>(v.ruw);
#pragma unroll 4
    for (int b = ug; b < v.r_nslots; b += (int)(blockDim.x >> 2)) ucol += rows[(size_t)b * (kCgMaxRecycle / 4) + uq];
    if (threadIdx.x < 2 * kCgMaxRecycle) sc[threadIdx.x] = v.rstate[(size_t)par * 2 * kCgMaxRecycle + threadIdx.x];
    if (threadIdx.x < kCgMaxRecycle * kCgMaxModes) sG[threadIdx.x] = v.dk ? v.rG[threadIdx.x] : 0.0;
  }
  if (v.dk && threadIdx.x < kCgMaxModes * kCgMaxModes) sE[threadIdx.x] = v.dsmall[threadIdx.x];
  const double dok = v.dk ? v.dsmall[72] : 0.0;
  const CgScal prev = v.scal[par];
  const double wn = v.delta_in_w ? v.w[v.n] : 0.0;
  const int done = v.st->done;
  if (done) return false;
  sg[g * 32 + l] = col;
  if (v.rk && !v.delta_in_w) {  // su[group][4 quarter + e]
    double* d = su + (threadIdx.x >> 2) * kCgMaxRecycle + 4 * (threadIdx.x & 3);
    d[0] = ucol.x;
    d[1] = ucol.y;
    d[2] = ucol.z;
    d[3] = ucol.w;
  }
  dl = wave_sum_dpp(dl);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) sg[(threadIdx.x >> 6) * 32 + 2] = dl;  // column 2 of groups 0..3: the waves' delta sums
  __syncthreads();
  if (threadIdx.x < 32) {
    double t = 0.0;
    const int ng = threadIdx.x == 2 ? (int)(blockDim.x >> 6) : 8;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < ng) t += sg[q * 32 + threadIdx.x];
    stot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double gamma = stot[0], delta = v.delta_in_w ? wn : stot[2];
    double y[kCgMaxModes];
#pragma unroll
    for (int i = 0; i < kCgMaxModes; ++i) y[i] = 0.0;
    if (v.dk && dok != 0.0) {
#pragma unroll
      for (int i = 0; i < kCgMaxModes; ++i)
        if (i < v.dk) {
          double a = 0.0;
#pragma unroll
          for (int j = 0; j < kCgMaxModes; ++j)
            if (j < v.dk) a += sE[i * kCgMaxModes + j] * stot[3 + j];
          y[i] = a;
          gamma -= a * stot[3 + kCgMaxModes + i];  // r.z_p = r.z - (W^T r).y
          delta -= a * stot[3 + i];                // z_p.w_p = z.w - y.((A W)^T z)
        }
    }
    double beta = 0.0, denom = delta;
    if (it > 0) {
      beta = prev.gamma > 0.0 ? gamma / prev.gamma : 0.0;
      denom = delta - beta * gamma / prev.alpha;
    }
    const bool ok = isfinite(denom) && denom > 0.0 && isfinite(gamma) && gamma >= 0.0;
    sb[0] = gamma;
    sb[1] = delta;
    sb[2] = ok ? gamma / denom : 0.0;
    sb[3] = ok ? beta : 0.0;
    sb[4] = ok ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < kCgMaxModes; ++i) sb[5 + i] = y[i];
  }
  if (v.rk) {
    // u_j . r by recurrence: s_new = w_p + beta s, r_new = r - alpha s_new, with u_j . w_p = u_j . w - sum_i y_i u_j . (A W_i)
    // (rounding drift in c_j only perturbs the preconditioner, not the system); thread j owns vector j
    __syncthreads();
    if (threadIdx.x < kCgMaxRecycle) {
      const int j = threadIdx.x;
      const bool ok = sb[4] != 0.0;
      const double al = sb[2], be = sb[3];
      double cn = 0.0, csn = 0.0;
      if (j < v.rk) {
        double uw = uwd;
        if (!v.delta_in_w) {
#pragma unroll 8
          for (int q = 0; q < 64; ++q) uw += su[q * kCgMaxRecycle + j];
        }
#pragma unroll
        for (int i = 0; i < kCgMaxModes; ++i) uw -= sb[5 + i] * sG[j * kCgMaxModes + i];
        csn = uw + be * sc[kCgMaxRecycle + j];
        cn = sc[j] - al * csn;
      }
      sa[j] = ok ? cn * v.rcoef[j] : 0.0;
      if (blockIdx.x == 0 && ok) {
        double* nxt = v.rstate + (size_t)((it + 1) & 1) * 2 * kCgMaxRecycle;
        nxt[j] = cn;
        nxt[kCgMaxRecycle + j] = csn;
      }
    }
  }
  __syncthreads();
  o.gamma = sb[0];
  o.delta = sb[1];
  o.alpha = sb[2];
  o.beta = sb[3];
  o.ok = sb[4] != 0.0;
#pragma unroll
  for (int i = 0; i < kCgMaxModes; ++i) o.y[i] = sb[5 + i];
#pragma unroll
  for (int j = 0; j < kCgMaxRecycle; ++j) o.a[j] = v.rk ? sa[j] : 0.0;
  __syncthreads();
  return true;
}

// Tail of k_cg_init* / k_cg_update*, called by all threads of every block after thread 0 has written the block's partial
// slots of parity `par`: the block that arrives LAST (ticket in st->pad; release fence before the ticket, acquire fence
// after it — agent scope, the per-XCD L2s are not coherent with each other) adds the r.r slots in slot order and records the
// status of iteration `next_it`: |r|^2, the iteration count, and `done` when |r|^2 <= tol^2 |b|^2.  The apply kernels then
// only look at st->done — until round 4 every workgroup of the first apply kernel re-reduced the slots itself (a dependent
// round trip, a block reduction and five barriers in front of a kernel that is three round trips long).
// `breakdown`: the step that led here had no positive curvature (identical in every block): the solve ends at `next_it - 1`.
__device__ __forceinline__ void cg_status_tail(const CgVec& v, int next_it, int par, bool breakdown, bool bad, double* smem /* >= 6 */) {
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    __threadfence();
    const int t = atomicAdd(&v.st->pad, 1);
    s_last = (t == (int)gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  double t[1] = {0.0};
  const double* vp = v.vpart + (size_t)par * kCgMaxBlocks * 2;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) t[0] += vp[2 * b + 1];
  block_sum<1>(t, smem);
  if (threadIdx.x == 0) {
    v.st->pad = 0;
    if (breakdown) {
      if (bad) v.st->bad = 1;
      v.st->done = 1;
      v.st->iters = next_it - 1;
      return;
    }
    const double bb = next_it == 0 ? t[0] : v.st->bb;
    const bool finite = isfinite(t[0]);
    if (next_it == 0) {
      v.st->bb = bb;
      v.st->bad = 0;
    }
    v.st->rr = t[0];
    v.st->iters = next_it;
    if (!finite) v.st->bad = 1;
    v.st->done = (!finite || t[0] <= v.tol2 * bb) ? 1 : 0;
  }
}

// Epilogue: this block's partials of the NEXT iterate (r.z | r.r | the 2k deflation products) in one block reduction; block
// 0 records the step's scalars and a breakdown.
__device__ __forceinline__ void cg_step_epilogue(const CgVec& v, int it, const CgStep& o, double rz, double rr,
                                                 double (&cd)[2 * kCgMaxModes], double* smem) {
  constexpr int KD = 2 * kCgMaxModes;
  double t[2 + KD];
  t[0] = rz;
  t[1] = rr;
#pragma unroll
  for (int j = 0; j < KD; ++j) t[2 + j] = cd[j];
  if (v.dk) block_sum_dpp<2 + KD>(t, smem); else block_sum_dpp<2>(reinterpret_cast<double(&)[2]>(t), smem);
  if (threadIdx.x == 0) {
    const int par = (it + 1) & 1;
    double* out = v.vpart + (size_t)par * kCgMaxBlocks * 2;
    out[blockIdx.x * 2] = t[0];
    out[blockIdx.x * 2 + 1] = t[1];
    if (v.dk) {
      double* oc = v.dcd + ((size_t)par * kCgMaxBlocks + blockIdx.x) * KD;
#pragma unroll
      for (int j = 0; j < KD; ++j) oc[j] = t[2 + j];
    }
    if (blockIdx.x == 0) {
      v.scal[par].gamma = o.gamma;
      v.scal[par].alpha = o.alpha;
      if (v.rhist && it < kCgHistCap) {  // Lanczos coefficients of this step (ritz.hpp)
        v.hist[2 * it] = o.gamma;
        v.hist[2 * it + 1] = o.alpha;
      }
    }
  }
  // gamma == 0 with a zero residual is plain convergence, anything else is a breakdown
  cg_status_tail(v, it + 1, (it + 1) & 1, !o.ok, !(o.gamma == 0.0 && isfinite(o.delta)), smem);
}
// z_p = z - W y, w_p = w - (A W) y at element o
__device__ __forceinline__ void cgd_project(const CgVec& v, long o, const double (&y)[kCgMaxModes], double& z, double& w) {
#pragma unroll
  for (int j = 0; j < kCgMaxModes; ++j)
    if (j < v.dk) {
      z -= y[j] * v.dW[(size_t)j * v.n + o];
      w -= y[j] * v.dAW[(size_t)j * v.n + o];
    }
}
__device__ __forceinline__ void cgd_acc_r(const CgVec& v, long o, double r, double (&cd)[2 * kCgMaxModes]) {
#pragma unroll
  for (int j = 0; j < kCgMaxModes; ++j)
    if (j < v.dk) cd[kCgMaxModes + j] += v.dW[(size_t)j * v.n + o] * r;
}
__device__ __forceinline__ void cgd_acc_z(const CgVec& v, long o, double z, double (&cd)[2 * kCgMaxModes]) {
#pragma unroll
  for (int j = 0; j < kCgMaxModes; ++j)
    if (j < v.dk) cd[j] += v.dAW[(size_t)j * v.n + o] * z;
}
template <int BS>
__device__ __forceinline__ void cg_block_init(const CgVec& v, long o, const double* __restrict__ m, double& rz,
                                              double& rr, double (&cd)[2 * kCgMaxModes],
                                              double* __restrict__ mir = nullptr) {
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
    cgd_acc_r(v, o + i, rb[i], cd);
    cgd_acc_z(v, o + i, zi, cd);
  }
}

// x = 0, r = b, z = M^-1 b, p = s = 0; partials of (r.z, r.r) into parity slot 0.
template <int PB, bool HAS_INTR, int IW = 8>
static __global__ void __launch_bounds__(kBlock) k_cg_init(CgVec v) {
  __shared__ double smem[4 * (2 + 2 * kCgMaxModes)];
  double acc[2] = {0.0, 0.0};
  double cd[2 * kCgMaxModes] = {};
  const int nblk = v.N + (HAS_INTR ? v.K : 0);
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += gridDim.x * blockDim.x) {
    if (b < v.N) {
      cg_block_init<PB>(v, (long)PB * b, v.minv + (long)PB * PB * b, acc[0], acc[1], cd,
                        v.zmir ? v.zmir + (long)b * v.zmir_stride + v.zmir_off : nullptr);
    } else if constexpr (HAS_INTR) {
      const int k = b - v.N;
      cg_block_init<IW>(v, (long)PB * v.N + (long)IW * k, v.minv + (long)PB * PB * v.N + (long)(IW * IW) * k, acc[0], acc[1], cd);
    }
  }
  {
    constexpr int KD = 2 * kCgMaxModes;
    double t[2 + KD];
    t[0] = acc[0];
    t[1] = acc[1];
#pragma unroll
    for (int j = 0; j < KD; ++j) t[2 + j] = cd[j];
    if (v.dk) block_sum<2 + KD>(t, smem); else block_sum<2>(reinterpret_cast<double(&)[2]>(t), smem);
    acc[0] = t[0];
    acc[1] = t[1];
    if (v.dk && threadIdx.x == 0) {
      double* oc = v.dcd + (size_t)blockIdx.x * KD;  // parity slot 0
#pragma unroll
      for (int j = 0; j < KD; ++j) oc[j] = t[2 + j];
    }
  }
  if (threadIdx.x == 0) {
    v.vpart[blockIdx.x * 2] = acc[0];
    v.vpart[blockIdx.x * 2 + 1] = acc[1];
    if (blockIdx.x == 0) {
      v.scal[0].gamma = 0.0;
      v.scal[0].alpha = 0.0;
    }
  }
  cg_status_tail(v, 0, 0, false, false, smem);
}

// Top of the first apply kernel of iteration `it`: true when the solve is finished (the caller returns).
__device__ __forceinline__ bool cg_converged(const CgVec& v, int it, double tol2, double* smem /* unused */) {
  (void)it;
  (void)tol2;
  (void)smem;
  if (v.probe) return false;  // forming A W: an operator application outside the iteration
  return v.st->done != 0;     // the vector kernel that produced this iterate has tested |r| (cg_status_tail, k_cg_*1)
}

// All loads of a block are issued BEFORE its first store: the vectors may alias as far as the compiler knows, so a load
// placed after a store waits for it — per element that was a chain of dependent round trips (36 us for 10 000 camera
// blocks with four deflated modes; the whole kernel is a handful of round trips now).
template <int BS, bool DEFL = true>
__device__ __forceinline__ void cg_block_update(const CgVec& v, long o, const double* __restrict__ m, double alpha,
                                                double beta, double& rz, double& rr,
                                                const double (&y)[kCgMaxModes], double (&cd)[2 * kCgMaxModes],
                                                double* __restrict__ mir = nullptr, const double* ra = nullptr, int it = 0) {
  double zi[BS], wi[BS], pi[BS], si[BS], xi[BS], ri[BS], mm[BS * BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    zi[i] = v.z[o + i];
    wi[i] = v.w[o + i];
    pi[i] = v.p[o + i];
    si[i] = v.s[o + i];
    xi[i] = v.x[o + i];
    ri[i] = v.r[o + i];
  }
#pragma unroll
  for (int i = 0; i < BS * BS; ++i) mm[i] = m[i];
  double rn[BS], zn[BS];
  if constexpr (!DEFL) {  // single-workgroup solves are never deflated (1 024-thread kernels: 128 registers)
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      pi[i] = zi[i] + beta * pi[i];
      si[i] = wi[i] + beta * si[i];
      xi[i] += alpha * pi[i];
      rn[i] = ri[i] - alpha * si[i];
      rr += rn[i] * rn[i];
    }
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < BS; ++j) t += mm[i * BS + j] * rn[j];
      zn[i] = t;
      rz += rn[i] * t;
    }
  } else if constexpr (BS <= 3) {
    // deflation: W and A W of the block's elements in registers, used for the projection and for the new dot products
    double Wv[BS][kCgMaxModes], AWv[BS][kCgMaxModes];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < kCgMaxModes; ++j) {
        Wv[i][j] = j < v.dk ? v.dW[(size_t)j * v.n + o + i] : 0.0;
        AWv[i][j] = j < v.dk ? v.dAW[(size_t)j * v.n + o + i] : 0.0;
      }
    // recycled Ritz vectors: the coarse part of the new z (loads issued here, before the first store below)
    double uz[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) uz[i] = 0.0;
    if (ra != nullptr && v.rk) {
#pragma unroll
      for (int j = 0; j < kCgMaxRecycle; ++j)
        if (j < v.rk) {
#pragma unroll
          for (int i = 0; i < BS; ++i) uz[i] += ra[j] * v.rU[(size_t)j * v.n + o + i];
        }
    }
#pragma unroll
    for (int i = 0; i < BS; ++i) {
#pragma unroll
      for (int j = 0; j < kCgMaxModes; ++j) {
        zi[i] -= y[j] * Wv[i][j];
        wi[i] -= y[j] * AWv[i][j];
      }
      pi[i] = zi[i] + beta * pi[i];
      si[i] = wi[i] + beta * si[i];
      xi[i] += alpha * pi[i];
      rn[i] = ri[i] - alpha * si[i];
      rr += rn[i] * rn[i];
#pragma unroll
      for (int j = 0; j < kCgMaxModes; ++j) cd[kCgMaxModes + j] += Wv[i][j] * rn[i];
    }
    if (v.rhist && it < kCgHistCap) {  // the projected z this step used: Lanczos vector `it` up to its norm
#pragma unroll
      for (int i = 0; i < BS; ++i) v.zhist[(size_t)it * v.n + o + i] = zi[i];
    }
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      double t = uz[i];
#pragma unroll
      for (int j = 0; j < BS; ++j) t += mm[i * BS + j] * rn[j];
      zn[i] = t;
      rz += rn[i] * t;
#pragma unroll
      for (int j = 0; j < kCgMaxModes; ++j) cd[j] += AWv[i][j] * t;
    }
  } else {
    // wide blocks (pose 6 / intrinsics 8 of the separate-block layout): the mode values are fetched per element
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      cgd_project(v, o + i, y, zi[i], wi[i]);
      pi[i] = zi[i] + beta * pi[i];
      si[i] = wi[i] + beta * si[i];
      xi[i] += alpha * pi[i];
      rn[i] = ri[i] - alpha * si[i];
      rr += rn[i] * rn[i];
      cgd_acc_r(v, o + i, rn[i], cd);
    }
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < BS; ++j) t += mm[i * BS + j] * rn[j];
      zn[i] = t;
      rz += rn[i] * t;
      cgd_acc_z(v, o + i, t, cd);
    }
  }
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    v.p[o + i] = pi[i];
    v.s[o + i] = si[i];
    v.x[o + i] = xi[i];
    v.r[o + i] = rn[i];
    v.z[o + i] = zn[i];
    if (mir) mir[i] = zn[i];
  }
}

// One CG iteration given w = A z (complete) and the delta partials.
template <int PB, bool HAS_INTR, int IW = 8>
static __global__ void __launch_bounds__(kBlock) k_cg_update(CgVec v, int it) {
  __shared__ double smem[kCgStepSmem];
  CgStep st;
  if (!cg_step_prologue(v, it, st, smem)) return;
  double acc[2] = {0.0, 0.0};
  double cd[2 * kCgMaxModes] = {};
  const int nblk = v.N + (HAS_INTR ? v.K : 0);
  if (st.ok) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += gridDim.x * blockDim.x) {
      if (b < v.N) {
        cg_block_update<PB>(v, (long)PB * b, v.minv + (long)PB * PB * b, st.alpha, st.beta, acc[0], acc[1], st.y, cd,
                            v.zmir ? v.zmir + (long)b * v.zmir_stride + v.zmir_off : nullptr, st.a, it);
      } else if constexpr (HAS_INTR) {
        const int k = b - v.N;
        cg_block_update<IW>(v, (long)PB * v.N + (long)IW * k, v.minv + (long)PB * PB * v.N + (long)(IW * IW) * k, st.alpha, st.beta,
                            acc[0], acc[1], st.y, cd);
      }
    }
  }
  cg_step_epilogue(v, it, st, acc[0], acc[1], cd, smem);
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
  double cd[2 * kCgMaxModes] = {};  // (single-workgroup solves are never deflated: v.dk == 0)
  for (int b = threadIdx.x; b < v.N; b += blockDim.x)
    cg_block_init<PB>(v, (long)PB * b, v.minv + (long)PB * PB * b, acc[0], acc[1], cd,
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
    const double y[kCgMaxModes] = {};
    double cd[2 * kCgMaxModes] = {};
    for (int b = threadIdx.x; b < v.N; b += blockDim.x)
      cg_block_update<PB, false>(v, (long)PB * b, v.minv + (long)PB * PB * b, alpha, beta, acc[0], acc[1], y, cd,
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
  __shared__ double smem[4 * (2 + 2 * kCgMaxModes)];
  __shared__ double sr[kJointCams][16];
  const int c = threadIdx.x >> 4, i = threadIdx.x & 15;
  double acc[2] = {0.0, 0.0};
  double cd[2 * kCgMaxModes] = {};
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
      cgd_acc_r(v, o, ri, cd);
      cgd_acc_z(v, o, zi, cd);
    }
  }
  {
    constexpr int KD = 2 * kCgMaxModes;
    double t[2 + KD];
    t[0] = acc[0];
    t[1] = acc[1];
#pragma unroll
    for (int j = 0; j < KD; ++j) t[2 + j] = cd[j];
    if (v.dk) block_sum<2 + KD>(t, smem); else block_sum<2>(reinterpret_cast<double(&)[2]>(t), smem);
    acc[0] = t[0];
    acc[1] = t[1];
    if (v.dk && threadIdx.x == 0) {
      double* oc = v.dcd + (size_t)blockIdx.x * KD;  // parity slot 0
#pragma unroll
      for (int j = 0; j < KD; ++j) oc[j] = t[2 + j];
    }
  }
  if (threadIdx.x == 0) {
    v.vpart[blockIdx.x * 2] = acc[0];
    v.vpart[blockIdx.x * 2 + 1] = acc[1];
    if (blockIdx.x == 0) {
      v.scal[0].gamma = 0.0;
      v.scal[0].alpha = 0.0;
    }
  }
  cg_status_tail(v, 0, 0, false, false, smem);
}

template <int PB>
static __global__ void __launch_bounds__(kBlock) k_cg_update_joint(CgVec v, int it) {
  constexpr int BJ = PB + 8;
  __shared__ double smem[kCgStepSmem];
  __shared__ double sr[kJointCams][16];
  CgStep st;
  if (!cg_step_prologue(v, it, st, smem)) return;
  const double alpha = st.alpha, beta = st.beta;
  double acc[2] = {0.0, 0.0};
  double cd[2 * kCgMaxModes] = {};
  const int c = threadIdx.x >> 4, i = threadIdx.x & 15;
  if (st.ok) {
    for (int n0 = blockIdx.x * kJointCams; n0 < v.N; n0 += gridDim.x * kJointCams) {
      const int n = n0 + c;
      const bool act = n < v.N && i < BJ;
      long o = 0;
      double ri = 0.0;
      double mrow[BJ], awv[kCgMaxModes];
      if (act) {
        o = cg_joint_index<PB>(v, n, i);
        // every load of this element first (see cg_block_update): vectors, its preconditioner row, its mode values
        double zo = v.z[o], wo = v.w[o];
        const double po = v.p[o], so = v.s[o], xo = v.x[o], ro = v.r[o];
        const double* m = v.minv_joint + n;
#pragma unroll
        for (int j = 0; j < BJ; ++j) mrow[j] = m[(size_t)(i * BJ + j) * v.N];
        double wv[kCgMaxModes];
#pragma unroll
        for (int j = 0; j < kCgMaxModes; ++j) {
          wv[j] = j < v.dk ? v.dW[(size_t)j * v.n + o] : 0.0;
          awv[j] = j < v.dk ? v.dAW[(size_t)j * v.n + o] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < kCgMaxModes; ++j) {
          zo -= st.y[j] * wv[j];
          wo -= st.y[j] * awv[j];
        }
        const double pi = zo + beta * po;
        const double si = wo + beta * so;
        ri = ro - alpha * si;
        v.p[o] = pi;
        v.s[o] = si;
        v.x[o] = xo + alpha * pi;
        v.r[o] = ri;
        acc[1] += ri * ri;
#pragma unroll
        for (int j = 0; j < kCgMaxModes; ++j) cd[kCgMaxModes + j] += wv[j] * ri;
      }
      __syncthreads();
      sr[c][i] = ri;
      __syncthreads();
      if (act) {
        double zi = 0.0;
#pragma unroll
        for (int j = 0; j < BJ; ++j) zi += mrow[j] * sr[c][j];
        v.z[o] = zi;
        cg_joint_mirror<PB>(v, n, i, zi);
        acc[0] += ri * zi;
#pragma unroll
        for (int j = 0; j < kCgMaxModes; ++j) cd[j] += awv[j] * zi;
      }
    }
  }
  cg_step_epilogue(v, it, st, acc[0], acc[1], cd, smem);
}

// Single-block reduction of the delta partials into w[n] (multi-rank: w[0..n] is all-reduced next) and, with recycled Ritz
// vectors in the preconditioner, of this rank's u_j . w slots into w[n + 1 + j]: the dot products travel with w.
static __global__ void __launch_bounds__(kBlock) k_cg_delta_to_w(CgVec v) {
  __shared__ double smem[4 + 1];
  __shared__ double su[64][kCgMaxRecycle];
  if (v.st->done) return;
  double d[1];
  reduce_partials<1>(v.dpart, v.nb_apply, d, smem);
  if (threadIdx.x == 0) v.w[v.n] = d[0];
  if (v.rk) {
    typedef double cg_d4v __attribute__((ext_vector_type(4)));
    cg_d4v ucol = {0.0, 0.0, 0.0, 0.0};
    const int uq = threadIdx.x & 3, ug = threadIdx.x >> 2;
    const cg_d4v* rows = reinterpret_cast<const cg_d4v*>(v.ruw);
#pragma unroll 4
    for (int b = ug; b < v.r_nslots; b += (int)(blockDim.x >> 2)) ucol += rows[(size_t)b * (kCgMaxRecycle / 4) + uq];
    su[ug][4 * uq] = ucol.x;
    su[ug][4 * uq + 1] = ucol.y;
    su[ug][4 * uq + 2] = ucol.z;
    su[ug][4 * uq + 3] = ucol.w;
    __syncthreads();
    if (threadIdx.x < kCgMaxRecycle) {
      double t = 0.0;
#pragma unroll 8
      for (int q = 0; q < 64; ++q) t += su[q][threadIdx.x];
      v.w[v.n + 1 + threadIdx.x] = t;
    }
  }
}


// Operator probes outside a solve (a solver applying A to vectors of its own, e.g. to build a coarse matrix): reset the status
// block the apply kernels look at, and set v.probe = 1 on the host view while probing.
static __global__ void k_cg_reset_status(CgVec v) {
  v.st->done = 0;
  v.st->iters = 0;
  v.st->bad = 0;
}

// ---- deflation: setup kernels (once per solve) -------------------------------------------------------------------
constexpr int kCgdBlocks = 128;                                   // grid of the Gram dot-product kernel
constexpr int kCgdGram = kCgMaxModes * (kCgMaxModes + 1);         // k x (k + 1) products W_i . (A W_j | b)

// Host view of a deflated solve.  W: the modes in the unknowns of the reduced system (zero in constant components).
struct CgDeflation {
  int k = 0;                   // modes in use (<= kCgMaxModes)
  const double* W = nullptr;   // [k][n]
  double* AW = nullptr;        // [k][n]
  int aw_ready = 0;            // the first aw_ready rows of AW were filled by the caller (closed-form products)
  double* b2 = nullptr;        // [n]   deflated right-hand side
  double* part = nullptr;      // [kCgdBlocks][kCgdGram] partial Gram products
  double* small = nullptr;     // [64] E^-1 | [8] y0 = E^-1 W^T b | [1] ok (1.0 / 0.0)
  double* cd = nullptr;        // [2][kCgMaxBlocks][2 * kCgMaxModes]  (CgVec::dcd)
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

// z := vec (vector and gather mirrors): the input of one operator application
template <int PB, bool JOINT>
static __global__ void __launch_bounds__(kBlock) k_cgd_set_z(CgVec v, const double* __restrict__ vec) {
  cgd_for_each<PB, JOINT>(v, [&](long o, int n, int i) {
    const double zi = vec[o];
    v.z[o] = zi;
    if constexpr (JOINT) {
      cg_joint_mirror<PB>(v, n, i, zi);
    } else {
      if (v.zmir && n >= 0) v.zmir[(long)n * v.zmir_stride + v.zmir_off + i] = zi;
    }
  });
}
static __global__ void __launch_bounds__(kBlock) k_cgd_copy(long n, const double* __restrict__ src, double* __restrict__ dst) {
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) dst[o] = src[o];
}

// partial Gram products of this block's slice, one row of E per blockIdx.y:
// part[block][i * (K + 1) + j] = W_i . (A W_j), j == K: W_i . b        (i = blockIdx.y)
template <int K>
static __global__ void __launch_bounds__(kBlock) k_cgd_gram_dots(CgVec v, CgDeflation d) {
  __shared__ double smem[4 * (K + 1)];
  const int i = blockIdx.y;
  double acc[K + 1] = {};
  const double* wi = d.W + (size_t)i * v.n;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < (long)v.n; o += (long)gridDim.x * blockDim.x) {
    const double a = wi[o];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] += a * d.AW[(size_t)j * v.n + o];
    acc[K] += a * v.b[o];
  }
  block_sum<K + 1>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j <= K; ++j) d.part[(size_t)blockIdx.x * kCgdGram + i * (K + 1) + j] = acc[j];
  }
}

// E = W^T A W (symmetrised), E^-1 by Gauss-Jordan with a positivity test, y0 = E^-1 W^T b.  One wave: lane j < 2K holds
// column j of [E | I], the K pivot steps run on registers with wave broadcasts.
template <int K>
static __global__ void __launch_bounds__(kBlock) k_cgd_gram_solve(CgDeflation d, int nparts) {
  __shared__ double sg[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  // column sums of the partial Gram products: lane l < K (K + 1) owns entry l; the four waves take a quarter of the
  // partial slots each (fixed order), wave 0 finishes
  {
    double gq = 0.0;
    if (lane < K * (K + 1))
      for (int b = q; b < nparts; b += 4) gq += d.part[(size_t)b * kCgdGram + lane];
    sg[q][lane] = gq;
  }
  __syncthreads();
  if (q != 0) return;
  const double g = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
  // a[r] = entry (r, lane) of [E_sym | I]
  double a[K];
  double rhs[K];  // W^T b, the same in every lane
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const double e_rc = __shfl(g, r * (K + 1) + (lane < K ? lane : 0), 64);
    const double e_cr = __shfl(g, (lane < K ? lane : 0) * (K + 1) + r, 64);
    a[r] = lane < K ? 0.5 * (e_rc + e_cr) : (lane - K == r ? 1.0 : 0.0);
    rhs[r] = __shfl(g, r * (K + 1) + K, 64);
  }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < K; ++c) {  // SPD: the pivots are the diagonal
    const double piv = __shfl(a[c], c, 64);
    if (!(piv > 0.0) || !isfinite(piv)) ok = false;
    const double ip = 1.0 / piv;
    const double rowc = a[c] * ip;  // scaled pivot row, this lane's column
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const double f = __shfl(a[r], c, 64);  // entry (r, c) before the step
      a[r] = r == c ? rowc : a[r] - f * rowc;
    }
  }
  // lanes K .. 2K-1 hold the columns of E^-1: lane K + j has (E^-1)[r][j] in a[r]
  double y0 = 0.0;  // lane K + j contributes (E^-1)[i][j] rhs[j] to y0[i]: sum over lanes per row
  double yv[K];
#pragma unroll
  for (int r = 0; r < K; ++r) {
    double t = (lane >= K && lane < 2 * K) ? a[r] * rhs[lane - K] : 0.0;
    t = wave_sum(t);
    yv[r] = __shfl(t, 0, 64);
    if (!isfinite(yv[r])) ok = false;
  }
  (void)y0;
  if (lane >= K && lane < 2 * K) {
#pragma unroll
    for (int r = 0; r < K; ++r) d.small[r * kCgMaxModes + (lane - K)] = ok ? a[r] : 0.0;
  }
  if (lane == 0) {
    for (int i = 0; i < kCgMaxModes; ++i) d.small[64 + i] = 0.0;
#pragma unroll
    for (int r = 0; r < K; ++r) d.small[64 + r] = ok ? yv[r] : 0.0;
    d.small[72] = ok ? 1.0 : 0.0;
  }
}

// b2 = b - A W y0
static __global__ void __launch_bounds__(kBlock) k_cgd_b2(CgVec v, CgDeflation d) {
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < (long)v.n; o += (long)gridDim.x * blockDim.x) {
    double t = v.b[o];
    for (int j = 0; j < d.k; ++j) t -= d.small[64 + j] * d.AW[(size_t)j * v.n + o];
    d.b2[o] = t;
  }
}
// x <- x + W y0
static __global__ void __launch_bounds__(kBlock) k_cgd_finish(CgVec v, CgDeflation d) {
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < (long)v.n; o += (long)gridDim.x * blockDim.x) {
    double xo = v.x[o];
    for (int j = 0; j < d.k; ++j) xo += d.small[64 + j] * d.W[(size_t)j * v.n + o];
    v.x[o] = xo;
  }
}

// ---- recycled Ritz vectors as an additive coarse space (CgRecycle; host side: ritz.hpp) --------------------------------
// The solves of one LM problem are a sequence of slowly changing systems, and what block-Jacobi leaves slow in one of them
// (global positioning at configs[3]: a tail of eigenvalues 0.01 ... 0.1 under a bulk in [0.2, 2]) is still slow in the next.
// A PCG solve computes the Lanczos coefficients of its own preconditioned operator for free; their small Ritz pairs
// (theta_j, u_j) are harvested after the solve (k_cgr_harvest forms u_j from the recorded z's) and the NEXT solves run with
//     M2^-1 = M^-1 + sum_j u_j u_j^T / theta_j
// — a fixed SPD preconditioner for the same system, the same right-hand side and the same stopping rule |r| <= tol |b|, so the
// solution is the same to the tolerance; (theta, u) are one or more linearisations old, which only costs preconditioner
// quality (CPU study on the C++ oracle, tools/exp_gp_ritz_recycle.py: 2 204 -> 1 646 iterations per GP solve of configs[3]).
// No launch is added to an iteration: u_j . r follows from the recurrence r_new = r - alpha (w_p + beta s) with the dot
// products u_j . w written as partial slots by the apply kernel that completes w (cg_step_prologue sums them), and the
// coarse part of the new z is added where k_cg_update forms M^-1 r_new anyway.
struct CgRecycle {
  int k = 0;                       // slots 0 .. k-1 may be in use
  double coef[kCgMaxRecycle] = {}; // 1 / theta_j, 0 for an empty slot
  double* U = nullptr;             // [kCgMaxRecycle][n]
  double* G = nullptr;             // [kCgMaxRecycle][kCgMaxModes]
  double* state = nullptr;         // [2][2 kCgMaxRecycle]
  double* uw = nullptr;            // [nslots][kCgMaxRecycle]: the apply kernels write these when v.rk != 0
  int nslots = 0;
  double* part = nullptr;          // [kCgrChunks][kCgMaxRecycle][1 + kCgMaxModes]
  bool record = false;             // record the Lanczos history of this solve
  double* zhist = nullptr;         // [kCgHistCap][n]
  double* hist = nullptr;          // [kCgHistCap][2]
};

// partial dot products of recycled vector j = blockIdx.y over chunk blockIdx.x of the unknowns:
// part[(chunk * kCgMaxRecycle + j) * 9 + (0: u_j . r0 | 1 + i: u_j . (A W_i))]
constexpr int kCgrChunks = 16;
static __global__ void __launch_bounds__(kBlock)
    k_cgr_init_dots(CgVec v, const double* __restrict__ AW, int dk, double* __restrict__ part) {
  __shared__ double smem[4 * (1 + kCgMaxModes)];
  const int j = blockIdx.y;
  double acc[1 + kCgMaxModes] = {};
  const double* u = v.rU + (size_t)j * v.n;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < (long)v.n; o += (long)gridDim.x * blockDim.x) {
    const double a = u[o];
    acc[0] += a * v.r[o];
#pragma unroll
    for (int i = 0; i < kCgMaxModes; ++i)
      if (i < dk) acc[1 + i] += a * AW[(size_t)i * v.n + o];
  }
  block_sum<1 + kCgMaxModes>(acc, smem);
  if (threadIdx.x == 0) {
    double* out = part + ((size_t)blockIdx.x * kCgMaxRecycle + j) * (1 + kCgMaxModes);
#pragma unroll
    for (int i = 0; i <= kCgMaxModes; ++i) out[i] = acc[i];
  }
}
// u_j . w of a COMPLETE w, for apply kernels that do not write the partial slots themselves (GP's plain camera-major sweep):
// one more launch per iteration, kCgrChunks slot rows.  Block (chunk, j) writes ruw[chunk][j]; columns j >= rk stay zero.
static __global__ void __launch_bounds__(kBlock) k_cgr_dots_w(CgVec v) {
  __shared__ double smem[4];
  if (v.st->done) return;
  const int j = blockIdx.y;
  double acc[1] = {0.0};
  const double* u = v.rU + (size_t)j * v.n;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < (long)v.n; o += (long)gridDim.x * blockDim.x) acc[0] += u[o] * v.w[o];
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) v.ruw[(size_t)blockIdx.x * kCgMaxRecycle + j] = acc[0];
}
// z0 += sum_j u_j (u_j . r0) / theta_j after k_cg_init (same grid, same element -> block map): z, its gather mirror, and the
// partials that depend on z (r.z and (A W)^T z of parity slot 0) are rewritten; block 0 starts the u_j . r recurrence.
template <int PB>
static __global__ void __launch_bounds__(kBlock) k_cgr_init_z(CgVec v, const double* __restrict__ part, int nchunks, double* __restrict__ G) {
  __shared__ double smem[4 * (1 + kCgMaxModes)];
  __shared__ double sa[kCgMaxRecycle];
  if (threadIdx.x < kCgMaxRecycle * (1 + kCgMaxModes)) {  // column (j, col) of the chunk partials
    const int j = threadIdx.x / (1 + kCgMaxModes), col = threadIdx.x % (1 + kCgMaxModes);
    double c = 0.0;
    if (j < v.rk) {
#pragma unroll 4
      for (int b = 0; b < nchunks; ++b) c += part[((size_t)b * kCgMaxRecycle + j) * (1 + kCgMaxModes) + col];
    }
    if (col == 0) {
      sa[j] = c * v.rcoef[j];
      if (blockIdx.x == 0) {
        v.rstate[j] = c;
        v.rstate[kCgMaxRecycle + j] = 0.0;
      }
    } else if (blockIdx.x == 0) {
      G[j * kCgMaxModes + (col - 1)] = c;
    }
  }
  __syncthreads();
  double a[kCgMaxRecycle];
#pragma unroll
  for (int j = 0; j < kCgMaxRecycle; ++j) a[j] = sa[j];
  double t[1 + kCgMaxModes] = {};
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < v.N; b += gridDim.x * blockDim.x) {
    const long o = (long)PB * b;
    double zi[PB], ri[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      zi[i] = v.z[o + i];
      ri[i] = v.r[o + i];
    }
#pragma unroll
    for (int j = 0; j < kCgMaxRecycle; ++j)
      if (j < v.rk) {
#pragma unroll
        for (int i = 0; i < PB; ++i) zi[i] += a[j] * v.rU[(size_t)j * v.n + o + i];
      }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      v.z[o + i] = zi[i];
      if (v.zmir) v.zmir[(long)b * v.zmir_stride + v.zmir_off + i] = zi[i];
      t[0] += ri[i] * zi[i];
#pragma unroll
      for (int j = 0; j < kCgMaxModes; ++j)
        if (j < v.dk) t[1 + j] += v.dAW[(size_t)j * v.n + o + i] * zi[i];
    }
  }
  block_sum<1 + kCgMaxModes>(t, smem);
  if (threadIdx.x == 0) {
    v.vpart[blockIdx.x * 2] = t[0];
    if (v.dk) {
      double* oc = v.dcd + (size_t)blockIdx.x * 2 * kCgMaxModes;  // parity slot 0, the (A W)^T z half
#pragma unroll
      for (int j = 0; j < kCgMaxModes; ++j) oc[j] = t[1 + j];
    }
  }
}
// u_e = sum_j coef[j][e] zhist[j]  ->  slot[e] of U   (coef from ritz_select, m recorded steps, knew <= 8 vectors)
struct CgrSlots {
  int s[8];
};
static __global__ void __launch_bounds__(kBlock)
    k_cgr_harvest(long n, int m, const double* __restrict__ zhist, const double* __restrict__ coef, int knew, CgrSlots slots,
                  double* __restrict__ U) {
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
    double acc[8] = {};
    for (int j = 0; j < m; ++j) {
      const double zj = zhist[(size_t)j * n + o];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += coef[j * 8 + e] * zj;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (e < knew) U[(size_t)slots.s[e] * n + o] = acc[e];
  }
}

// Host driver.  `apply(it)` must enqueue the kernels computing w = A z and the delta partials (its first kernel calls
// cg_converged); it is also responsible for timing its dominant kernel (ctx->prof.begin(s, id, it)).
//   defl   optional: deflate these modes (multi-block vector kernels only; small single-workgroup solves run plain).
//          The k operator applications that form A W are counted in the returned iteration count.
//   finished  optional, out: set when the solve ended by itself (convergence / breakdown) rather than at max_iter.
//   hint   optional, in/out: the iteration count of the previous solve of this sequence.  The host enqueues iterations
//          ahead of the device and reads the status back first where the previous solve ended, then every third
//          iteration — so a converged solve leaves a couple of early-exit launches behind instead of a chunk of them.
struct CgNoPostZ {
  void operator()(int) const {}
};
// post_z(par): optional hook enqueued after every kernel that produces a new z = M^-1 r (k_cg_init*: par = 0;
// k_cg_update* of iteration it: par = (it + 1) & 1) — a second-level preconditioner adds its coarse correction to z (and to
// the gather mirrors) there and rewrites the r.z partials of parity slot `par` (gp.hip: GpCoarse).
template <int PB, bool HAS_INTR, int IW = 8, typename Apply, typename PostZ = CgNoPostZ>
inline long cg_solve(gsfm_ctx* ctx, CgVec& v, double tol, int max_iter, Apply&& apply, const CgDeflation* defl = nullptr,
                     int* hint = nullptr, PostZ&& post_z = PostZ(), bool* finished = nullptr, const CgRecycle* rcy = nullptr) {
  hipStream_t s = ctx->stream;
  const bool multi = ctx->comm.world > 1;
  v.delta_in_w = multi ? 1 : 0;
  const bool joint = HAS_INTR && IW == 8 && v.joint_map != nullptr;
  v.tol2 = tol * tol;
  v.single = (!HAS_INTR && v.N <= kCgSingleMaxBlocks) ? 1 : 0;
  v.probe = 0;
  v.dk = 0;
  v.rk = 0;
  v.rhist = 0;
  auto init = [&]() {
    if constexpr (HAS_INTR) {
      if (joint) hipLaunchKernelGGL((k_cg_init_joint<PB>), dim3(v.nb_update), dim3(kBlock), 0, s, v);
    } else {
      if (v.single) hipLaunchKernelGGL((k_cg_init1<PB>), dim3(1), dim3(kCgSingleThreads), 0, s, v);
    }
    if (!joint && !v.single) hipLaunchKernelGGL((k_cg_init<PB, HAS_INTR, IW>), dim3(v.nb_update), dim3(kBlock), 0, s, v);
  };
  auto apply_all = [&](int it) {  // w = A z complete on every rank
    apply(it);
    if (multi) {
      hipLaunchKernelGGL(k_cg_delta_to_w, dim3(1), dim3(kBlock), 0, s, v);
      allreduce_sum(ctx, v.w, (size_t)v.n + 1 + (v.rk ? kCgMaxRecycle : 0));  // (w needs n + 1 + kCgMaxRecycle entries then)
    }
  };
  GSFM_HIP_CHECK(hipMemsetAsync(v.st, 0, sizeof(CgStatus), s));  // (the ticket of cg_status_tail starts at zero)
  const bool deflate = defl != nullptr && defl->k > 0 && !v.single;
  const double* b_caller = v.b;
  const int gvec = std::min(256, grid_for((size_t)std::max((long)v.n, (long)v.N * 16), kBlock));
  if (deflate) {
    init();  // (resets the status block the apply kernels look at)
    // k_cg_init's status tail has already tested |b|: a zero or non-finite right-hand side raises `done` here, and the
    // sweeps of the probes below look at st->done directly — they would return early and leave A W stale for the Gram
    // kernels.  The probes run with the flag cleared; the second init() (on b2) raises it again where it belongs.
    if (defl->aw_ready < defl->k) hipLaunchKernelGGL(k_cg_reset_status, dim3(1), dim3(1), 0, s, v);
    v.probe = 1;
    for (int j = defl->aw_ready; j < defl->k; ++j) {  // A W_j: the operator reads z (and its mirrors), writes w
      if (joint)
        hipLaunchKernelGGL((k_cgd_set_z<PB, true>), dim3(gvec), dim3(kBlock), 0, s, v, defl->W + (size_t)j * v.n);
      else
        hipLaunchKernelGGL((k_cgd_set_z<PB, false>), dim3(gvec), dim3(kBlock), 0, s, v, defl->W + (size_t)j * v.n);
      apply_all(0);
      hipLaunchKernelGGL(k_cgd_copy, dim3(gvec), dim3(kBlock), 0, s, (long)v.n, (const double*)v.w, defl->AW + (size_t)j * v.n);
    }
    v.probe = 0;
    const int gdot = std::min(kCgdBlocks, grid_for((size_t)v.n, kBlock));
    if (defl->k == 4) {
      hipLaunchKernelGGL((k_cgd_gram_dots<4>), dim3(gdot, 4), dim3(kBlock), 0, s, v, *defl);
      hipLaunchKernelGGL((k_cgd_gram_solve<4>), dim3(1), dim3(kBlock), 0, s, *defl, gdot);
    } else if (defl->k == 7) {
      hipLaunchKernelGGL((k_cgd_gram_dots<7>), dim3(gdot, 7), dim3(kBlock), 0, s, v, *defl);
      hipLaunchKernelGGL((k_cgd_gram_solve<7>), dim3(1), dim3(kBlock), 0, s, *defl, gdot);
    } else {
      throw StatusError(GSFM_ERR_INVALID_ARGUMENT, "cg_solve: 4 or 7 deflated modes");
    }
    hipLaunchKernelGGL(k_cgd_b2, dim3(gvec), dim3(kBlock), 0, s, v, *defl);
    v.b = defl->b2;
    v.dk = defl->k;
    v.dW = defl->W;
    v.dAW = defl->AW;
    v.dsmall = defl->small;
    v.dcd = defl->cd;
  }
  init();
  // recycled Ritz vectors (3 x 3 camera blocks, multi-block vector kernels, one rank): coarse part of z0, start of the recurrences
  bool recycle = false;
  if constexpr (PB == 3 && !HAS_INTR) recycle = rcy != nullptr && !v.single;
  if constexpr (PB == 3 && !HAS_INTR) if (recycle) {
    v.rhist = rcy->record ? 1 : 0;
    v.zhist = rcy->zhist;
    v.hist = rcy->hist;
    if (rcy->k > 0) {
      v.rk = rcy->k;
      v.rU = rcy->U;
      v.rG = rcy->G;
      v.rstate = rcy->state;
      v.ruw = rcy->uw;
      v.r_nslots = rcy->nslots;
      for (int j = 0; j < kCgMaxRecycle; ++j) v.rcoef[j] = rcy->coef[j];
      hipLaunchKernelGGL(k_cgr_init_dots, dim3(kCgrChunks, v.rk), dim3(kBlock), 0, s, v, deflate ? (const double*)defl->AW : nullptr,
                         deflate ? defl->k : 0, rcy->part);
      hipLaunchKernelGGL((k_cgr_init_z<PB>), dim3(v.nb_update), dim3(kBlock), 0, s, v, (const double*)rcy->part, kCgrChunks, rcy->G);
    }
  }
  post_z(0);
  CgStatus* h = reinterpret_cast<CgStatus*>(ctx->h_pinned + 400);
  long iters = max_iter;
  // convergence after p iterations is recorded by the vector kernel of iteration p - 1: read back after p enqueued iterations
  int next_poll = (hint && *hint > 0) ? *hint : 8;
  if (ctx->comm.host_fn) next_poll = 1;  // (host-staged transport: the stream is drained every iteration anyway)
  for (int it = 0; it < max_iter; ++it) {
    apply_all(it);
    if constexpr (HAS_INTR) {
      if (joint) hipLaunchKernelGGL((k_cg_update_joint<PB>), dim3(v.nb_update), dim3(kBlock), 0, s, v, it);
    } else {
      if (v.single) hipLaunchKernelGGL((k_cg_update1<PB>), dim3(1), dim3(kCgSingleThreads), 0, s, v, it);
    }
    if (!joint && !v.single) hipLaunchKernelGGL((k_cg_update<PB, HAS_INTR, IW>), dim3(v.nb_update), dim3(kBlock), 0, s, v, it);
    post_z((it + 1) & 1);
    if (it + 1 >= next_poll || it == max_iter - 1) {
      GSFM_HIP_CHECK(hipMemcpyAsync(h, v.st, sizeof(CgStatus), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipStreamSynchronize(s));
      GSFM_HIP_CHECK(hipGetLastError());
      comm_check(ctx);
      ctx->prof.harvest(h->done ? h->iters : 0x7fffffff);  // launches of iterations >= iters found `done` and left at once
      if (h->done) {
        iters = h->iters;
        if (finished) *finished = true;  // (converged or broke down — not "ran into max_iter")
        break;
      }
      next_poll = it + 1 + (ctx->comm.host_fn ? 1 : 3);
    }
  }
  if (hint) *hint = (int)iters;
  v.rk = 0;
  v.rhist = 0;
  ctx->stats[GSFM_STAT_PCG_SOLVES]++;
  ctx->stats[GSFM_STAT_PCG_ITERATIONS] += iters;
  if (v.single) ctx->stats[GSFM_STAT_PCG_SINGLE_WORKGROUP]++;
  if (joint) ctx->stats[GSFM_STAT_PCG_JOINT_BLOCKS]++;
  if (deflate) {
    ctx->stats[GSFM_STAT_PCG_DEFLATED]++;
    if (defl->aw_ready == defl->k) ctx->stats[GSFM_STAT_PCG_CLOSED_FORM_AW]++;
    hipLaunchKernelGGL(k_cgd_finish, dim3(gvec), dim3(kBlock), 0, s, v, *defl);
    v.b = b_caller;
    v.dk = 0;
    iters += defl->k - defl->aw_ready;  // the operator applications that formed A W
  }
  return iters;
}

}  // namespace gsfm
