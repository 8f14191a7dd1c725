// ba.hip — global bundle adjustment on MI355X (gfx950).
//
// Replaces BundleAdjuster::Solve (glomap/estimators/bundle_adjustment.cc:11-106) for trivial rigs:
//   residual   colmap::ReprojErrorCostFunctor<CameraModel> (ba.cc:135-146): r = ImgFromCam(params, R(q) X + t) - obs
//   unknowns   frame pose (EigenQuaternionManifold tangent 3 + translation 3), point (3),
//              shared intrinsics blocks (principal point frozen unless optimize_principal_point)
//   constants  first frame q and t (ba.cc:261-266); all q when !optimize_rotations (gm.cc:208)
//   loss       Huber(1 px) (bundle_adjustment.h:30)
//   solver     Ceres LM + SPARSE_SCHUR  ->  lm.hpp + 3x3 point elimination + implicit-Schur PCG on the
//              reduced camera system (6 per frame + 8 per intrinsics block), block-Jacobi preconditioned.
//
// Nothing per-observation is stored except the robust weight: Jacobians are analytic and cheap, so
// every sweep recomputes them from (R, t) of the frame (96 B, L2-resident gather), the point
// (24 B, track-contiguous) and the intrinsics.  With a = R X, h = Jx^T g:
//   J_rot v = Jx (2 v x a),  J_rot^T g = 2 a x h,  J_trn = Jx,  J_pt = Jx R,  J_intr = Jp.
//
// Reduced-system vector layout: [6 per frame (rot, trn) | 8 per intrinsics block]; constant or
// non-existent entries keep a zero Jacobian column (their step is exactly 0).
//
// Data layout in HBM (f64 unless noted; observations track-major):
//   pt_offset[P+1] i64, obs_cam[M] i32, obs_xy[M][2], cam_intr[N] i32, intr_model[K] i32     inputs
//   q[N][4], t[N][3], camR[N][9], X[P][3], par[K][8] + candidates                              state
//   wrob[M]                                                                                    per observation
//   hinv[P][6], ept[P][3], ptdiag[P][3], ptjs[P][3], used[P] u8                                per track
//   diag, js, dvec, grad, gred, rhs [6N+8K];  spose[N][21], sintr[K][36], minv[36N+64K]        reduced system
#include "camera.hpp"
#include "cgvec.hpp"
#include "lm.hpp"

namespace gsfm {
namespace {

struct BaParams {
  int N, K;
  long P, M;
  const long* off;
  const int* cam;
  const double* xy;
  const int* cam_intr;
  const int* intr_model;
  const unsigned char* used;
  const unsigned char* intr_free;  // [K] bit j set = params[j] is optimised
  int fixed_cam;
  int opt_rot, opt_trn, opt_pts;
  double huber_a;
  double lm_lo, lm_hi;
};

__device__ __forceinline__ int sym6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }
__device__ __forceinline__ int sym8(int i, int j) { return i * 8 - (i * (i - 1)) / 2 + (j - i); }

// Per-thread accumulator of W values keyed by a small integer (the intrinsics block id): values
// are added up in registers while consecutive observations share the key and only flushed with
// atomics when the key changes.  finish() reduces across the block first when every thread ended
// on the same key — the shared-intrinsics case, where per-observation atomics on one address
// would serialise the whole sweep.
template <int W>
struct RunAcc {
  int key = -1;
  double v[W];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = 0.0;
  }
  __device__ __forceinline__ void flush(double* __restrict__ base) {
    if (key >= 0) {
#pragma unroll
      for (int i = 0; i < W; ++i)
        if (v[i] != 0.0) unsafeAtomicAdd(base + (long)key * W + i, v[i]);
    }
    clear();
    key = -1;
  }
  __device__ __forceinline__ void select(int k, double* __restrict__ base) {
    if (k != key) {
      flush(base);
      key = k;
    }
  }
  // all threads of the block must call this
  __device__ __forceinline__ void finish(double* __restrict__ base, double* smem /* >= 4*W doubles */, int* skey) {
    if (threadIdx.x == 0) *skey = -1;
    __syncthreads();
    if (key >= 0) atomicMax(skey, key);
    __syncthreads();
    const int ref = *skey;
    const int same = __syncthreads_and(key == ref || key < 0);
    if (same && ref >= 0) {
      block_sum<W>(v, smem);
      if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < W; ++i)
          if (v[i] != 0.0) unsafeAtomicAdd(base + (long)ref * W + i, v[i]);
      }
    } else {
      flush(base);
    }
  }
};

struct ObsJac {
  double Jpose[2][6];  // [rot | trn], masked
  double Jpt[2][3];    // masked by opt_pts
};

__device__ __forceinline__ void build_jac(const BaParams& g, int n, const double* __restrict__ R9, const ObsGeom& o,
                                          ObsJac& J) {
  const bool rf = g.opt_rot && n != g.fixed_cam;
  const bool tf = g.opt_trn && n != g.fixed_cam;
  const V3 a = o.a;
  // C = -2 [a]x
  const double C[3][3] = {{0.0, 2.0 * a.z, -2.0 * a.y}, {-2.0 * a.z, 0.0, 2.0 * a.x}, {2.0 * a.y, -2.0 * a.x, 0.0}};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J.Jpose[r][j] = rf ? (o.Jx[r][0] * C[0][j] + o.Jx[r][1] * C[1][j] + o.Jx[r][2] * C[2][j]) : 0.0;
      J.Jpose[r][3 + j] = tf ? o.Jx[r][j] : 0.0;
      J.Jpt[r][j] = g.opt_pts ? (o.Jx[r][0] * R9[j] + o.Jx[r][1] * R9[3 + j] + o.Jx[r][2] * R9[6 + j]) : 0.0;
    }
  }
}

__device__ __forceinline__ void mask_intr(unsigned char bits, double (&Jp)[2][8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (!((bits >> j) & 1)) {
      Jp[0][j] = 0.0;
      Jp[1][j] = 0.0;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
    k_ba_cam_prepare(int N, const double* __restrict__ q, double* __restrict__ camR) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const double w = q[4 * n], x = q[4 * n + 1], y = q[4 * n + 2], z = q[4 * n + 3];
    double* R = camR + 9 * (long)n;
    R[0] = 1 - 2 * (y * y + z * z);
    R[1] = 2 * (x * y - w * z);
    R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);
    R[4] = 1 - 2 * (x * x + z * z);
    R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);
    R[7] = 2 * (y * z + w * x);
    R[8] = 1 - 2 * (x * x + y * y);
  }
}

// ---- linearize ---------------------------------------------------------------------------------
// cost, robust weights, gradient, squared column norms.  part[block][2] = {cost, max |g_pt|}.
__global__ void __launch_bounds__(kBlock)
    k_ba_linearize(BaParams g, const double* __restrict__ camR, const double* __restrict__ t,
                   const double* __restrict__ X, const double* __restrict__ par, double* __restrict__ wrob,
                   double* __restrict__ ptdiag, double* __restrict__ diag, double* __restrict__ grad,
                   double* __restrict__ intr_acc /* [K][16]: diag 8 | grad 8 */, double* __restrict__ part) {
  __shared__ double smem[4 * 16];
  __shared__ int skey;
  double cost = 0.0, gmax = 0.0;
  RunAcc<16> ia;
  ia.clear();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < g.P; p += (long)gridDim.x * blockDim.x) {
    if (!g.used[p]) continue;
    const V3 Xp = ld3(X + 3 * p);
    double pd[3] = {0, 0, 0}, pg[3] = {0, 0, 0};
    for (long k = g.off[p]; k < g.off[p + 1]; ++k) {
      const int n = g.cam[k];
      const int ik = g.cam_intr[n];
      const double* R9 = camR + 9 * (long)n;
      ObsGeom o;
      obs_geom(R9, t + 3 * (long)n, Xp, g.intr_model[ik], par + 8 * (long)ik, o);
      const double r0 = o.valid ? o.px - g.xy[2 * k] : 0.0;
      const double r1 = o.valid ? o.py - g.xy[2 * k + 1] : 0.0;
      double rho, w;
      huber(g.huber_a, 1.0, r0 * r0 + r1 * r1, rho, w);
      if (!o.valid) w = 0.0;
      wrob[k] = w;
      cost += 0.5 * rho;
      ObsJac J;
      build_jac(g, n, R9, o, J);
      mask_intr(g.intr_free[ik], o.Jp);
      const double g0 = w * r0, g1 = w * r1;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double dj = w * (J.Jpose[0][j] * J.Jpose[0][j] + J.Jpose[1][j] * J.Jpose[1][j]);
        const double gj = J.Jpose[0][j] * g0 + J.Jpose[1][j] * g1;
        if (dj != 0.0) unsafeAtomicAdd(diag + 6 * (long)n + j, dj);
        if (gj != 0.0) unsafeAtomicAdd(grad + 6 * (long)n + j, gj);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        pd[j] += w * (J.Jpt[0][j] * J.Jpt[0][j] + J.Jpt[1][j] * J.Jpt[1][j]);
        pg[j] += J.Jpt[0][j] * g0 + J.Jpt[1][j] * g1;
      }
      ia.select(ik, intr_acc);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ia.v[j] += w * (o.Jp[0][j] * o.Jp[0][j] + o.Jp[1][j] * o.Jp[1][j]);
        ia.v[8 + j] += o.Jp[0][j] * g0 + o.Jp[1][j] * g1;
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      ptdiag[3 * p + j] = pd[j];
      gmax = fmax(gmax, fabs(pg[j]));
    }
  }
  ia.finish(intr_acc, smem, &skey);
  double v[1] = {cost};
  block_sum<1>(v, smem);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_down(gmax, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[4 + (threadIdx.x >> 6)] = gmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = v[0];
    part[blockIdx.x * 2 + 1] = fmax(fmax(smem[4], smem[5]), fmax(smem[6], smem[7]));
  }
}

// scatter intr_acc [K][16] into the reduced-vector layout: diag/grad [6N + 8k + j]
__global__ void __launch_bounds__(kBlock)
    k_ba_intr_unpack16(int N, int K, const double* __restrict__ intr_acc, double* __restrict__ diag,
                       double* __restrict__ grad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 8 * K; i += gridDim.x * blockDim.x) {
    const int k = i / 8, j = i % 8;
    diag[6 * (long)N + i] = intr_acc[16 * (long)k + j];
    grad[6 * (long)N + i] = intr_acc[16 * (long)k + 8 + j];
  }
}

// out[0] = sum part[.][0]; out[1] = max(part[.][1], max |vec|)
__global__ void __launch_bounds__(kBlock)
    k_ba_finalize_lin(const double* __restrict__ part, int nblocks, const double* __restrict__ vec, int nvec,
                      double* __restrict__ out) {
  __shared__ double smem[8];
  double cost = 0.0, gmax = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    cost += part[2 * b];
    gmax = fmax(gmax, part[2 * b + 1]);
  }
  for (int i = threadIdx.x; i < nvec; i += blockDim.x) gmax = fmax(gmax, fabs(vec[i]));
  double v[1] = {cost};
  block_sum<1>(v, smem);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_down(gmax, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[4 + (threadIdx.x >> 6)] = gmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = v[0];
    out[1] = fmax(fmax(smem[4], smem[5]), fmax(smem[6], smem[7]));
  }
}

// js = 1 / (1 + sqrt(diag)) (Ceres jacobi_scaling, fixed at the initial point)
__global__ void __launch_bounds__(kBlock)
    k_ba_jacobi(long n, int enabled, const double* __restrict__ diag, double* __restrict__ js) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    js[i] = enabled ? 1.0 / (1.0 + sqrt(diag[i])) : 1.0;
}

__device__ __forceinline__ double lm_damping(double h, double js, double radius, double lo, double hi) {
  const double j2 = js * js;
  return fmin(fmax(j2 * h, lo), hi) / (radius * j2);
}

// ---- build (radius dependent) ---------------------------------------------------------------------
// Per track: H_pp (+ damping) -> inverse, e = H_pp^-1 g_p; per observation the reduced gradient
// J_a^T w (r - J_pt e) and the diagonal Schur blocks J_a^T W_k J_a, W_k = w (I - w J_pt H_pp^-1 J_pt^T).
__global__ void __launch_bounds__(kBlock)
    k_ba_build(BaParams g, double radius, const double* __restrict__ camR, const double* __restrict__ t,
               const double* __restrict__ X, const double* __restrict__ par, const double* __restrict__ wrob,
               const double* __restrict__ ptdiag, const double* __restrict__ ptjs, double* __restrict__ hinv,
               double* __restrict__ ept, double* __restrict__ gred, double* __restrict__ spose,
               double* __restrict__ intr_acc /* [K][44]: gred 8 | S 36 */) {
  __shared__ double smem[4 * 44];
  __shared__ int skey;
  RunAcc<44> ia;
  ia.clear();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < g.P; p += (long)gridDim.x * blockDim.x) {
    if (!g.used[p]) continue;
    const V3 Xp = ld3(X + 3 * p);
    S3 H{0, 0, 0, 0, 0, 0};
    V3 gp{0, 0, 0};
    for (long k = g.off[p]; k < g.off[p + 1]; ++k) {
      const int n = g.cam[k];
      const int ik = g.cam_intr[n];
      const double* R9 = camR + 9 * (long)n;
      ObsGeom o;
      obs_geom(R9, t + 3 * (long)n, Xp, g.intr_model[ik], par + 8 * (long)ik, o);
      const double w = wrob[k];
      const double r0 = o.valid ? o.px - g.xy[2 * k] : 0.0;
      const double r1 = o.valid ? o.py - g.xy[2 * k + 1] : 0.0;
      ObsJac J;
      build_jac(g, n, R9, o, J);
      H.xx += w * (J.Jpt[0][0] * J.Jpt[0][0] + J.Jpt[1][0] * J.Jpt[1][0]);
      H.xy += w * (J.Jpt[0][0] * J.Jpt[0][1] + J.Jpt[1][0] * J.Jpt[1][1]);
      H.xz += w * (J.Jpt[0][0] * J.Jpt[0][2] + J.Jpt[1][0] * J.Jpt[1][2]);
      H.yy += w * (J.Jpt[0][1] * J.Jpt[0][1] + J.Jpt[1][1] * J.Jpt[1][1]);
      H.yz += w * (J.Jpt[0][1] * J.Jpt[0][2] + J.Jpt[1][1] * J.Jpt[1][2]);
      H.zz += w * (J.Jpt[0][2] * J.Jpt[0][2] + J.Jpt[1][2] * J.Jpt[1][2]);
      gp.x += w * (J.Jpt[0][0] * r0 + J.Jpt[1][0] * r1);
      gp.y += w * (J.Jpt[0][1] * r0 + J.Jpt[1][1] * r1);
      gp.z += w * (J.Jpt[0][2] * r0 + J.Jpt[1][2] * r1);
    }
    S3 Hi{0, 0, 0, 0, 0, 0};
    V3 e{0, 0, 0};
    if (g.opt_pts) {
      H.xx += lm_damping(ptdiag[3 * p], ptjs[3 * p], radius, g.lm_lo, g.lm_hi);
      H.yy += lm_damping(ptdiag[3 * p + 1], ptjs[3 * p + 1], radius, g.lm_lo, g.lm_hi);
      H.zz += lm_damping(ptdiag[3 * p + 2], ptjs[3 * p + 2], radius, g.lm_lo, g.lm_hi);
      Hi = inv3(H);
      e = mul(Hi, gp);
    }
    double* hp = hinv + 6 * p;
    hp[0] = Hi.xx; hp[1] = Hi.xy; hp[2] = Hi.xz; hp[3] = Hi.yy; hp[4] = Hi.yz; hp[5] = Hi.zz;
    st3(ept + 3 * p, e);
    for (long k = g.off[p]; k < g.off[p + 1]; ++k) {
      const int n = g.cam[k];
      const int ik = g.cam_intr[n];
      const double* R9 = camR + 9 * (long)n;
      ObsGeom o;
      obs_geom(R9, t + 3 * (long)n, Xp, g.intr_model[ik], par + 8 * (long)ik, o);
      const double w = wrob[k];
      if (w == 0.0) continue;
      const double r0 = o.px - g.xy[2 * k], r1 = o.py - g.xy[2 * k + 1];
      ObsJac J;
      build_jac(g, n, R9, o, J);
      mask_intr(g.intr_free[ik], o.Jp);
      // r - J_pt e
      const double q0 = r0 - (J.Jpt[0][0] * e.x + J.Jpt[0][1] * e.y + J.Jpt[0][2] * e.z);
      const double q1 = r1 - (J.Jpt[1][0] * e.x + J.Jpt[1][1] * e.y + J.Jpt[1][2] * e.z);
      // T = J_pt Hinv J_pt^T (2x2 sym), W = w (I - w T)
      const V3 h0 = mul(Hi, V3{J.Jpt[0][0], J.Jpt[0][1], J.Jpt[0][2]});
      const V3 h1 = mul(Hi, V3{J.Jpt[1][0], J.Jpt[1][1], J.Jpt[1][2]});
      const double T00 = J.Jpt[0][0] * h0.x + J.Jpt[0][1] * h0.y + J.Jpt[0][2] * h0.z;
      const double T01 = J.Jpt[0][0] * h1.x + J.Jpt[0][1] * h1.y + J.Jpt[0][2] * h1.z;
      const double T11 = J.Jpt[1][0] * h1.x + J.Jpt[1][1] * h1.y + J.Jpt[1][2] * h1.z;
      const double W00 = w * (1.0 - w * T00), W01 = -w * w * T01, W11 = w * (1.0 - w * T11);
      double* gr = gred + 6 * (long)n;
      double* sp = spose + 21 * (long)n;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double gi = w * (J.Jpose[0][i] * q0 + J.Jpose[1][i] * q1);
        if (gi != 0.0) unsafeAtomicAdd(gr + i, gi);
        const double a0 = W00 * J.Jpose[0][i] + W01 * J.Jpose[1][i];
        const double a1 = W01 * J.Jpose[0][i] + W11 * J.Jpose[1][i];
#pragma unroll
        for (int j = i; j < 6; ++j) {
          const double sij = a0 * J.Jpose[0][j] + a1 * J.Jpose[1][j];
          if (sij != 0.0) unsafeAtomicAdd(sp + sym6(i, j), sij);
        }
      }
      ia.select(ik, intr_acc);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ia.v[i] += w * (o.Jp[0][i] * q0 + o.Jp[1][i] * q1);
        const double a0 = W00 * o.Jp[0][i] + W01 * o.Jp[1][i];
        const double a1 = W01 * o.Jp[0][i] + W11 * o.Jp[1][i];
#pragma unroll
        for (int j = i; j < 8; ++j) ia.v[8 + sym8(i, j)] += a0 * o.Jp[0][j] + a1 * o.Jp[1][j];
      }
    }
  }
  ia.finish(intr_acc, smem, &skey);
}

// One thread per block of the block-Jacobi preconditioner: damping, rhs = -g', dense inverse.
__global__ void __launch_bounds__(kBlock)
    k_ba_blocks_finalize(int N, int K, double radius, double lo, double hi, const double* __restrict__ diag,
                         const double* __restrict__ js, const double* __restrict__ gred,
                         const double* __restrict__ spose, const double* __restrict__ intr_acc,
                         double* __restrict__ dvec, double* __restrict__ rhs, double* __restrict__ minv) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < N + K; b += gridDim.x * blockDim.x) {
    double A[64];
    if (b < N) {
      const long o = 6 * (long)b;
      const double* sp = spose + 21 * (long)b;
      for (int i = 0; i < 6; ++i) {
        const double D = lm_damping(diag[o + i], js[o + i], radius, lo, hi);
        dvec[o + i] = D;
        rhs[o + i] = -gred[o + i];
        for (int j = i; j < 6; ++j) {
          const double v = sp[sym6(i, j)] + (i == j ? D : 0.0);
          A[i * 6 + j] = v;
          A[j * 6 + i] = v;
        }
      }
      spd_inverse<8>(A, 6);
      double* m = minv + 36 * (long)b;
      for (int i = 0; i < 36; ++i) m[i] = A[i];
    } else {
      const int k = b - N;
      const long o = 6 * (long)N + 8 * (long)k;
      const double* acc = intr_acc + 44 * (long)k;
      for (int i = 0; i < 8; ++i) {
        const double D = lm_damping(diag[o + i], js[o + i], radius, lo, hi);
        dvec[o + i] = D;
        rhs[o + i] = -acc[i];
        for (int j = i; j < 8; ++j) {
          const double v = acc[8 + sym8(i, j)] + (i == j ? D : 0.0);
          A[i * 8 + j] = v;
          A[j * 8 + i] = v;
        }
      }
      spd_inverse<8>(A, 8);
      double* m = minv + 36 * (long)N + 64 * (long)k;
      for (int i = 0; i < 64; ++i) m[i] = A[i];
    }
  }
}

__global__ void __launch_bounds__(kBlock)
    k_ba_block_desc(int N, int K, int* __restrict__ elem_blk, int* __restrict__ blk_start,
                    int* __restrict__ blk_size, int* __restrict__ blk_moff) {
  const int n = 6 * N + 8 * K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < 6 * N) {
      const int b = i / 6;
      elem_blk[i] = b;
      if (i % 6 == 0) {
        blk_start[b] = i;
        blk_size[b] = 6;
        blk_moff[b] = 36 * b;
      }
    } else {
      const int k = (i - 6 * N) / 8;
      elem_blk[i] = N + k;
      if ((i - 6 * N) % 8 == 0) {
        blk_start[N + k] = i;
        blk_size[N + k] = 8;
        blk_moff[N + k] = 36 * N + 64 * k;
      }
    }
  }
}

// ---- the hot kernel: y += (H_aa - H_ap H_pp^-1 H_pa) v over this rank's tracks ------------------
// One thread per track, Jacobians recomputed on the fly.  Algorithmic bytes per launch
// (SURVEY.md §8d, K-BA-res, matrix-free): 28 M + 392 N + 120 P + 32 K.
__global__ void __launch_bounds__(kBlock)
    k_ba_schur_matvec(BaParams g, const double* __restrict__ camR, const double* __restrict__ t,
                      const double* __restrict__ X, const double* __restrict__ par,
                      const double* __restrict__ wrob, const double* __restrict__ hinv,
                      const double* __restrict__ v, double* __restrict__ y) {
  __shared__ double smem[4 * 8];
  __shared__ int skey;
  RunAcc<8> ia;
  ia.clear();
  double* yintr = y + 6 * (long)g.N;
  const double* vintr = v + 6 * (long)g.N;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < g.P; p += (long)gridDim.x * blockDim.x) {
    if (!g.used[p]) continue;
    const V3 Xp = ld3(X + 3 * p);
    const long k0 = g.off[p], k1 = g.off[p + 1];
    V3 acc{0, 0, 0};
    for (long k = k0; k < k1; ++k) {
      const double w = wrob[k];
      if (w == 0.0) continue;
      const int n = g.cam[k];
      const int ik = g.cam_intr[n];
      const double* R9 = camR + 9 * (long)n;
      ObsGeom o;
      obs_geom(R9, t + 3 * (long)n, Xp, g.intr_model[ik], par + 8 * (long)ik, o);
      const bool rf = g.opt_rot && n != g.fixed_cam, tf = g.opt_trn && n != g.fixed_cam;
      const double* vp = v + 6 * (long)n;
      V3 om{0, 0, 0};
      if (rf) om = 2.0 * cross(V3{vp[0], vp[1], vp[2]}, o.a);
      if (tf) om = om + V3{vp[3], vp[4], vp[5]};
      double u0, u1;
      jx_mul(o.Jx, om, u0, u1);
      const unsigned char bits = g.intr_free[ik];
      const double* vi = vintr + 8 * (long)ik;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((bits >> j) & 1) {
          u0 += o.Jp[0][j] * vi[j];
          u1 += o.Jp[1][j] * vi[j];
        }
      }
      acc = acc + RT_mul(R9, jxT_mul(o.Jx, w * u0, w * u1));  // J_pt^T w u
    }
    V3 tp{0, 0, 0};
    if (g.opt_pts) {
      const double* hp = hinv + 6 * p;
      tp = mul(S3{hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]}, acc);
    }
    for (long k = k0; k < k1; ++k) {
      const double w = wrob[k];
      if (w == 0.0) continue;
      const int n = g.cam[k];
      const int ik = g.cam_intr[n];
      const double* R9 = camR + 9 * (long)n;
      ObsGeom o;
      obs_geom(R9, t + 3 * (long)n, Xp, g.intr_model[ik], par + 8 * (long)ik, o);
      const bool rf = g.opt_rot && n != g.fixed_cam, tf = g.opt_trn && n != g.fixed_cam;
      const double* vp = v + 6 * (long)n;
      V3 om{0, 0, 0};
      if (rf) om = 2.0 * cross(V3{vp[0], vp[1], vp[2]}, o.a);
      if (tf) om = om + V3{vp[3], vp[4], vp[5]};
      om = om - R_mul(R9, tp);  // ... - J_pt t_p
      double u0, u1;
      jx_mul(o.Jx, om, u0, u1);
      const unsigned char bits = g.intr_free[ik];
      const double* vi = vintr + 8 * (long)ik;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((bits >> j) & 1) {
          u0 += o.Jp[0][j] * vi[j];
          u1 += o.Jp[1][j] * vi[j];
        }
      }
      const double g0 = w * u0, g1 = w * u1;
      const V3 h = jxT_mul(o.Jx, g0, g1);
      double* yp = y + 6 * (long)n;
      if (rf) atomic_add3(yp, 2.0 * cross(o.a, h));
      if (tf) atomic_add3(yp + 3, h);
      if (bits) {
        ia.select(ik, yintr);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if ((bits >> j) & 1) ia.v[j] += o.Jp[0][j] * g0 + o.Jp[1][j] * g1;
      }
    }
  }
  ia.finish(yintr, smem, &skey);
}

// ---- back-substitution, model cost change, candidate points ------------------------------------
// part[block][3] = {model_cost_change, |dX|^2, |X|^2}
__global__ void __launch_bounds__(kBlock)
    k_ba_backsub(BaParams g, const double* __restrict__ camR, const double* __restrict__ t,
                 const double* __restrict__ X, const double* __restrict__ par, const double* __restrict__ wrob,
                 const double* __restrict__ hinv, const double* __restrict__ ept, const double* __restrict__ dv,
                 double* __restrict__ Xn, double* __restrict__ part) {
  __shared__ double smem[4 * 3];
  double acc3[3] = {0, 0, 0};
  const double* dintr = dv + 6 * (long)g.N;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < g.P; p += (long)gridDim.x * blockDim.x) {
    const V3 Xp = ld3(X + 3 * p);
    if (!g.used[p]) {
      st3(Xn + 3 * p, Xp);
      continue;
    }
    const long k0 = g.off[p], k1 = g.off[p + 1];
    V3 dX{0, 0, 0};
    for (int pass = 0; pass < 2; ++pass) {
      V3 acc{0, 0, 0};
      for (long k = k0; k < k1; ++k) {
        const double w = wrob[k];
        if (w == 0.0) continue;
        const int n = g.cam[k];
        const int ik = g.cam_intr[n];
        const double* R9 = camR + 9 * (long)n;
        ObsGeom o;
        obs_geom(R9, t + 3 * (long)n, Xp, g.intr_model[ik], par + 8 * (long)ik, o);
        const bool rf = g.opt_rot && n != g.fixed_cam, tf = g.opt_trn && n != g.fixed_cam;
        const double* vp = dv + 6 * (long)n;
        V3 om{0, 0, 0};
        if (rf) om = 2.0 * cross(V3{vp[0], vp[1], vp[2]}, o.a);
        if (tf) om = om + V3{vp[3], vp[4], vp[5]};
        if (pass == 1 && g.opt_pts) om = om + R_mul(R9, dX);
        double u0, u1;
        jx_mul(o.Jx, om, u0, u1);
        const unsigned char bits = g.intr_free[ik];
        const double* vi = dintr + 8 * (long)ik;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if ((bits >> j) & 1) {
            u0 += o.Jp[0][j] * vi[j];
            u1 += o.Jp[1][j] * vi[j];
          }
        }
        if (pass == 0) {
          acc = acc + RT_mul(R9, jxT_mul(o.Jx, w * u0, w * u1));
        } else {
          const double r0 = o.px - g.xy[2 * k], r1 = o.py - g.xy[2 * k + 1];
          acc3[0] -= w * (u0 * r0 + u1 * r1 + 0.5 * (u0 * u0 + u1 * u1));
        }
      }
      if (pass == 0 && g.opt_pts) {
        const double* hp = hinv + 6 * p;
        dX = V3{0, 0, 0} - ld3(ept + 3 * p) - mul(S3{hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]}, acc);
      }
    }
    st3(Xn + 3 * p, Xp + dX);
    acc3[1] += dot(dX, dX);
    acc3[2] += dot(Xp, Xp);
  }
  block_sum<3>(acc3, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) part[blockIdx.x * 3 + k] = acc3[k];
  }
}

// Candidate poses / intrinsics (single block): q' = [cos|d|, sin|d| d/|d|] * q (EigenQuaternionManifold),
// t' = t + dt, par' = par + dpar.  out = {|step|^2, |x|^2, #non-finite}.
__global__ void __launch_bounds__(kCgThreads)
    k_ba_param_update(int N, int K, const double* __restrict__ q, const double* __restrict__ t,
                      const double* __restrict__ par, const double* __restrict__ dv, double* __restrict__ qn,
                      double* __restrict__ tn, double* __restrict__ parn, double* __restrict__ out) {
  __shared__ double smem[17];
  double st = 0.0, xn = 0.0, bad = 0.0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const double* d = dv + 6 * (long)n;
    const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double s, c;
    sincos(th, &s, &c);
    const double k = th > 0.0 ? s / th : 1.0;
    const Quat qd{c, k * d[0], k * d[1], k * d[2]};
    const Quat q0{q[4 * n], q[4 * n + 1], q[4 * n + 2], q[4 * n + 3]};
    const Quat q1 = qmul(qd, q0);
    qn[4 * n] = q1.w; qn[4 * n + 1] = q1.x; qn[4 * n + 2] = q1.y; qn[4 * n + 3] = q1.z;
    st += (q1.w - q0.w) * (q1.w - q0.w) + (q1.x - q0.x) * (q1.x - q0.x) + (q1.y - q0.y) * (q1.y - q0.y) +
          (q1.z - q0.z) * (q1.z - q0.z);
    xn += q0.w * q0.w + q0.x * q0.x + q0.y * q0.y + q0.z * q0.z;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double tj = t[3 * n + j];
      tn[3 * n + j] = tj + d[3 + j];
      st += d[3 + j] * d[3 + j];
      xn += tj * tj;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) bad += isfinite(d[j]) ? 0.0 : 1.0;
  }
  for (int i = threadIdx.x; i < 8 * K; i += blockDim.x) {
    const double d = dv[6 * (long)N + i];
    parn[i] = par[i] + d;
    st += d * d;
    xn += par[i] * par[i];
    bad += isfinite(d) ? 0.0 : 1.0;
  }
  st = block_dot_1024(st, smem);
  xn = block_dot_1024(xn, smem);
  bad = block_dot_1024(bad, smem);
  if (threadIdx.x == 0) {
    out[0] = st;
    out[1] = xn;
    out[2] = bad;
  }
}

__global__ void __launch_bounds__(kBlock)
    k_ba_cost(BaParams g, const double* __restrict__ camR, const double* __restrict__ t,
              const double* __restrict__ X, const double* __restrict__ par, double* __restrict__ part) {
  __shared__ double smem[4];
  double v[1] = {0.0};
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < g.P; p += (long)gridDim.x * blockDim.x) {
    if (!g.used[p]) continue;
    const V3 Xp = ld3(X + 3 * p);
    for (long k = g.off[p]; k < g.off[p + 1]; ++k) {
      const int n = g.cam[k];
      const int ik = g.cam_intr[n];
      ObsGeom o;
      obs_geom(camR + 9 * (long)n, t + 3 * (long)n, Xp, g.intr_model[ik], par + 8 * (long)ik, o);
      if (!o.valid) continue;
      const double r0 = o.px - g.xy[2 * k], r1 = o.py - g.xy[2 * k + 1];
      double rho, w;
      huber(g.huber_a, 1.0, r0 * r0 + r1 * r1, rho, w);
      v[0] += 0.5 * rho;
    }
  }
  block_sum<1>(v, smem);
  if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}

template <int K>
__global__ void __launch_bounds__(kBlock)
    k_ba_sum_partials(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
  __shared__ double smem[4 * K + K];
  double tot[K];
  reduce_partials<K>(part, nblocks, tot, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = tot[k];
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct BaWs {
  DevBuf<long> off;
  DevBuf<int> cam, cam_intr, intr_model, elem_blk, blk_start, blk_size, blk_moff;
  DevBuf<unsigned char> used, intr_free;
  DevBuf<double> xy, q, qn, t, tn, camR, camRn, X, Xn, par, parn, wrob, hinv, ept, ptdiag, ptjs, diag, js, dvec,
      grad, gred, rhs, spose, iacc16, iacc44, minv, cg_x, cg_r, cg_z, cg_p, cg_y, part, scal;
  DevBuf<CgState> cg;
  static void destroy(void* p) { delete static_cast<BaWs*>(p); }
};

BaWs* ba_ws(gsfm_ctx* ctx) {
  if (!ctx->ba_ws) {
    ctx->ba_ws = new BaWs();
    ctx->ba_ws_free = &BaWs::destroy;
  }
  return static_cast<BaWs*>(ctx->ba_ws);
}

int num_params_of(int model) {
  switch (model) {
    case GSFM_CAMERA_SIMPLE_PINHOLE: return 3;
    case GSFM_CAMERA_PINHOLE: return 4;
    case GSFM_CAMERA_SIMPLE_RADIAL: return 4;
    case GSFM_CAMERA_RADIAL: return 5;
    case GSFM_CAMERA_OPENCV: return 8;
    default: return -1;
  }
}
unsigned pp_mask_of(int model) {
  switch (model) {
    case GSFM_CAMERA_PINHOLE:
    case GSFM_CAMERA_OPENCV: return (1u << 2) | (1u << 3);
    default: return (1u << 1) | (1u << 2);
  }
}

class BaSolver final : public LmProblem {
 public:
  BaSolver(gsfm_ctx* ctx, const gsfm_ba_options& opt) : ctx_(ctx), ws_(ba_ws(ctx)), opt_(opt) {}

  void setup(const gsfm_ba_problem* prob, const double* cam_q, const double* cam_t, const double* pt_xyz,
             const double* intr_params) {
    BaWs* ws = ws_;
    hipStream_t s = ctx_->stream;
    const int mem = prob->mem;
    N_ = prob->num_cams;
    K_ = prob->num_intr;
    P_ = prob->num_pts;
    M_ = prob->num_obs;
    n_ = 6 * N_ + 8 * K_;
    GSFM_REQUIRE(N_ > 0 && K_ > 0 && P_ >= 0 && M_ >= 0, "BA: bad sizes");
    GSFM_REQUIRE(prob->fixed_cam >= -1 && prob->fixed_cam < N_, "BA: fixed_cam out of range");
    std::vector<long> h_off;
    to_host(ctx_, h_off, reinterpret_cast<const long*>(prob->pt_offset), (size_t)P_ + 1, mem);
    GSFM_REQUIRE(h_off[0] == 0 && h_off[P_] == M_, "BA: pt_offset must start at 0 and end at num_obs");
    std::vector<unsigned char> h_used(P_);
    m_used_ = 0;
    for (long p = 0; p < P_; ++p) {
      const long len = h_off[p + 1] - h_off[p];
      GSFM_REQUIRE(len >= 0, "BA: pt_offset must be non-decreasing");
      h_used[p] = len >= opt_.min_num_view_per_track ? 1 : 0;  // ba.cc:122
      if (h_used[p]) m_used_ += len;
    }
    std::vector<int> h_model, h_ci;
    to_host(ctx_, h_model, prob->intr_model, (size_t)K_, mem);
    to_host(ctx_, h_ci, prob->cam_intr, (size_t)N_, mem);
    std::vector<unsigned char> h_free(K_);
    for (int k = 0; k < K_; ++k) {
      const int np = num_params_of(h_model[k]);
      if (np < 0) throw StatusError(GSFM_ERR_UNSUPPORTED, "BA: camera model not supported");
      unsigned bits = 0;
      // ba.cc:273-293: SubsetManifold on the principal point / constant block / everything free
      if (opt_.optimize_intrinsics || opt_.optimize_principal_point) {
        bits = (1u << np) - 1u;
        if (opt_.optimize_intrinsics && !opt_.optimize_principal_point) bits &= ~pp_mask_of(h_model[k]);
      }
      h_free[k] = (unsigned char)bits;
    }
    for (int n = 0; n < N_; ++n) GSFM_REQUIRE(h_ci[n] >= 0 && h_ci[n] < K_, "BA: cam_intr out of range");
    copy_in(ctx_, ws->off.ensure(P_ + 1), reinterpret_cast<const long*>(prob->pt_offset), (size_t)P_ + 1, mem);
    copy_in(ctx_, ws->cam.ensure(M_ + 1), prob->obs_cam, (size_t)M_, mem);
    copy_in(ctx_, ws->xy.ensure(2 * (size_t)M_ + 2), prob->obs_xy, 2 * (size_t)M_, mem);
    copy_in(ctx_, ws->cam_intr.ensure(N_), prob->cam_intr, (size_t)N_, mem);
    copy_in(ctx_, ws->intr_model.ensure(K_), prob->intr_model, (size_t)K_, mem);
    copy_in(ctx_, ws->q.ensure(4 * (size_t)N_), cam_q, 4 * (size_t)N_, mem);
    copy_in(ctx_, ws->t.ensure(3 * (size_t)N_), cam_t, 3 * (size_t)N_, mem);
    copy_in(ctx_, ws->X.ensure(3 * (size_t)P_ + 3), pt_xyz, 3 * (size_t)P_, mem);
    copy_in(ctx_, ws->par.ensure(8 * (size_t)K_), intr_params, 8 * (size_t)K_, mem);
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->used.ensure(P_ + 1), h_used.data(), (size_t)P_, hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->intr_free.ensure(K_), h_free.data(), (size_t)K_, hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    ws->qn.ensure(4 * (size_t)N_);
    ws->tn.ensure(3 * (size_t)N_);
    ws->camR.ensure(9 * (size_t)N_);
    ws->camRn.ensure(9 * (size_t)N_);
    ws->Xn.ensure(3 * (size_t)P_ + 3);
    ws->parn.ensure(8 * (size_t)K_);
    ws->wrob.ensure(M_ + 1);
    ws->hinv.ensure(6 * (size_t)P_ + 6);
    for (DevBuf<double>* b : {&ws->ept, &ws->ptdiag, &ws->ptjs}) b->ensure(3 * (size_t)P_ + 3);
    for (DevBuf<double>* b : {&ws->diag, &ws->js, &ws->dvec, &ws->grad, &ws->gred, &ws->rhs, &ws->cg_x, &ws->cg_r,
                              &ws->cg_z, &ws->cg_p, &ws->cg_y})
      b->ensure(n_);
    ws->spose.ensure(21 * (size_t)N_);
    ws->iacc16.ensure(16 * (size_t)K_);
    ws->iacc44.ensure(44 * (size_t)K_);
    ws->minv.ensure(36 * (size_t)N_ + 64 * (size_t)K_);
    ws->part.ensure(kMaxBlocks * 4);
    ws->scal.ensure(64);
    ws->cg.ensure(1);
    ws->elem_blk.ensure(n_);
    for (DevBuf<int>* b : {&ws->blk_start, &ws->blk_size, &ws->blk_moff}) b->ensure(N_ + K_);
    gridP_ = grid_for(P_, kBlock);
    gridN_ = grid_for(N_, kBlock);
    hipLaunchKernelGGL(k_ba_block_desc, dim3(grid_for(n_, kBlock)), dim3(kBlock), 0, s, N_, K_, ws->elem_blk.get(),
                       ws->blk_start.get(), ws->blk_size.get(), ws->blk_moff.get());
    g_.N = N_;
    g_.K = K_;
    g_.P = P_;
    g_.M = M_;
    g_.off = ws->off.get();
    g_.cam = ws->cam.get();
    g_.xy = ws->xy.get();
    g_.cam_intr = ws->cam_intr.get();
    g_.intr_model = ws->intr_model.get();
    g_.used = ws->used.get();
    g_.intr_free = ws->intr_free.get();
    g_.fixed_cam = prob->fixed_cam;
    g_.opt_rot = opt_.optimize_rotations ? 1 : 0;
    g_.opt_trn = opt_.optimize_translation ? 1 : 0;
    g_.opt_pts = opt_.optimize_points ? 1 : 0;
    g_.huber_a = opt_.thres_loss_function;
    g_.lm_lo = opt_.lm.min_lm_diagonal;
    g_.lm_hi = opt_.lm.max_lm_diagonal;
    q_ = ws->q.get(); qn_ = ws->qn.get();
    t_ = ws->t.get(); tn_ = ws->tn.get();
    R_ = ws->camR.get(); Rn_ = ws->camRn.get();
    X_ = ws->X.get(); Xn_ = ws->Xn.get();
    par_ = ws->par.get(); parn_ = ws->parn.get();
    hipLaunchKernelGGL(k_ba_cam_prepare, dim3(gridN_), dim3(kBlock), 0, s, N_, q_, R_);
    bj_ = BlockJacobi{ws->elem_blk.get(), ws->blk_start.get(), ws->blk_size.get(), ws->blk_moff.get(), ws->minv.get()};
  }

  long used_observations() const { return m_used_; }

  double linearize(double* grad_max_norm) override {
    BaWs* ws = ws_;
    hipStream_t s = ctx_->stream;
    GSFM_HIP_CHECK(hipMemsetAsync(ws->diag.get(), 0, (size_t)n_ * sizeof(double), s));
    GSFM_HIP_CHECK(hipMemsetAsync(ws->grad.get(), 0, (size_t)n_ * sizeof(double), s));
    GSFM_HIP_CHECK(hipMemsetAsync(ws->iacc16.get(), 0, 16 * (size_t)K_ * sizeof(double), s));
    hipLaunchKernelGGL(k_ba_linearize, dim3(gridP_), dim3(kBlock), 0, s, g_, R_, t_, X_, par_, ws->wrob.get(),
                       ws->ptdiag.get(), ws->diag.get(), ws->grad.get(), ws->iacc16.get(), ws->part.get());
    hipLaunchKernelGGL(k_ba_intr_unpack16, dim3(grid_for(8 * (size_t)K_, kBlock)), dim3(kBlock), 0, s, N_, K_,
                       ws->iacc16.get(), ws->diag.get(), ws->grad.get());
    if (ctx_->comm.world > 1) {
      allreduce_sum(ctx_, ws->diag.get(), n_);
      allreduce_sum(ctx_, ws->grad.get(), n_);
    }
    hipLaunchKernelGGL(k_ba_finalize_lin, dim3(1), dim3(kBlock), 0, s, ws->part.get(), gridP_, ws->grad.get(), n_,
                       ws->scal.get());
    if (ctx_->comm.world > 1) {
      allreduce_sum(ctx_, ws->scal.get(), 1);
      GSFM_NCCL_CHECK(ncclAllReduce(ws->scal.get() + 1, ws->scal.get() + 1, 1, ncclDouble, ncclMax, ctx_->comm.nccl, s));
    }
    GSFM_HIP_CHECK(hipMemcpyAsync(ctx_->h_pinned + 300, ws->scal.get(), 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    GSFM_HIP_CHECK(hipGetLastError());
    *grad_max_norm = ctx_->h_pinned[301];
    return ctx_->h_pinned[300];
  }

  void set_jacobi_scaling(bool enabled) override {
    BaWs* ws = ws_;
    hipStream_t s = ctx_->stream;
    hipLaunchKernelGGL(k_ba_jacobi, dim3(grid_for(n_, kBlock)), dim3(kBlock), 0, s, (long)n_, enabled ? 1 : 0,
                       ws->diag.get(), ws->js.get());
    hipLaunchKernelGGL(k_ba_jacobi, dim3(grid_for(3 * (size_t)P_, kBlock)), dim3(kBlock), 0, s, 3 * P_,
                       enabled ? 1 : 0, ws->ptdiag.get(), ws->ptjs.get());
  }

  bool step(double radius, double* model_change, double* cand_cost, double* step_norm, double* x_norm,
            long* linear_iterations) override {
    BaWs* ws = ws_;
    hipStream_t s = ctx_->stream;
    const bool multi = ctx_->comm.world > 1;
    GSFM_HIP_CHECK(hipMemsetAsync(ws->gred.get(), 0, (size_t)n_ * sizeof(double), s));
    GSFM_HIP_CHECK(hipMemsetAsync(ws->spose.get(), 0, 21 * (size_t)N_ * sizeof(double), s));
    GSFM_HIP_CHECK(hipMemsetAsync(ws->iacc44.get(), 0, 44 * (size_t)K_ * sizeof(double), s));
    hipLaunchKernelGGL(k_ba_build, dim3(gridP_), dim3(kBlock), 0, s, g_, radius, R_, t_, X_, par_, ws->wrob.get(),
                       ws->ptdiag.get(), ws->ptjs.get(), ws->hinv.get(), ws->ept.get(), ws->gred.get(),
                       ws->spose.get(), ws->iacc44.get());
    if (multi) {
      allreduce_sum(ctx_, ws->gred.get(), 6 * (size_t)N_);
      allreduce_sum(ctx_, ws->spose.get(), 21 * (size_t)N_);
      allreduce_sum(ctx_, ws->iacc44.get(), 44 * (size_t)K_);
    }
    hipLaunchKernelGGL(k_ba_blocks_finalize, dim3(grid_for(N_ + K_, kBlock)), dim3(kBlock), 0, s, N_, K_, radius,
                       g_.lm_lo, g_.lm_hi, ws->diag.get(), ws->js.get(), ws->gred.get(), ws->spose.get(),
                       ws->iacc44.get(), ws->dvec.get(), ws->rhs.get(), ws->minv.get());
    *linear_iterations = pcg();
    hipLaunchKernelGGL(k_ba_backsub, dim3(gridP_), dim3(kBlock), 0, s, g_, R_, t_, X_, par_, ws->wrob.get(),
                       ws->hinv.get(), ws->ept.get(), ws->cg_x.get(), Xn_, ws->part.get());
    hipLaunchKernelGGL((k_ba_sum_partials<3>), dim3(1), dim3(kBlock), 0, s, ws->part.get(), gridP_, ws->scal.get());
    hipLaunchKernelGGL(k_ba_param_update, dim3(1), dim3(kCgThreads), 0, s, N_, K_, q_, t_, par_, ws->cg_x.get(), qn_,
                       tn_, parn_, ws->scal.get() + 3);
    hipLaunchKernelGGL(k_ba_cam_prepare, dim3(gridN_), dim3(kBlock), 0, s, N_, qn_, Rn_);
    hipLaunchKernelGGL(k_ba_cost, dim3(gridP_), dim3(kBlock), 0, s, g_, Rn_, tn_, Xn_, parn_, ws->part.get());
    hipLaunchKernelGGL((k_ba_sum_partials<1>), dim3(1), dim3(kBlock), 0, s, ws->part.get(), gridP_, ws->scal.get() + 6);
    if (multi) {
      allreduce_sum(ctx_, ws->scal.get(), 3);
      allreduce_sum(ctx_, ws->scal.get() + 6, 1);
    }
    double h[7];
    GSFM_HIP_CHECK(hipMemcpyAsync(ctx_->h_pinned + 256, ws->scal.get(), 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    GSFM_HIP_CHECK(hipGetLastError());
    std::memcpy(h, ctx_->h_pinned + 256, sizeof(h));
    *model_change = h[0];
    *step_norm = std::sqrt(h[1] + h[3]);
    *x_norm = std::sqrt(h[2] + h[4]);
    *cand_cost = h[6];
    return h[5] == 0.0 && std::isfinite(h[0]) && std::isfinite(h[1]) && std::isfinite(h[6]);
  }

  void accept() override {
    std::swap(q_, qn_);
    std::swap(t_, tn_);
    std::swap(R_, Rn_);
    std::swap(X_, Xn_);
    std::swap(par_, parn_);
  }

  void write_back(const gsfm_ba_problem* prob, double* cam_q, double* cam_t, double* pt_xyz, double* intr) {
    copy_out(ctx_, cam_q, q_, 4 * (size_t)N_, prob->mem);
    copy_out(ctx_, cam_t, t_, 3 * (size_t)N_, prob->mem);
    copy_out(ctx_, pt_xyz, X_, 3 * (size_t)P_, prob->mem);
    copy_out(ctx_, intr, par_, 8 * (size_t)K_, prob->mem);
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx_->stream));
  }

 private:
  long pcg() {
    BaWs* ws = ws_;
    hipStream_t s = ctx_->stream;
    const bool multi = ctx_->comm.world > 1;
    const double yscale = ctx_->comm.rank == 0 ? 1.0 : 0.0;
    const double tol = opt_.lm.pcg_relative_tolerance;
    hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(kCgThreads), 0, s, n_, ws->rhs.get(), ws->cg_x.get(), ws->cg_r.get(),
                       ws->cg_z.get(), ws->cg_p.get(), ws->cg_y.get(), ws->dvec.get(), bj_, ws->cg.get(), yscale);
    CgState* h = reinterpret_cast<CgState*>(ctx_->h_pinned + 400);
    const int chunk = 8;
    const int max_iter = opt_.lm.pcg_max_iterations;
    for (int it = 0; it < max_iter; ++it) {
      const bool timed = ctx_->prof.begin(s, GSFM_KERNEL_BA_SCHUR);
      hipLaunchKernelGGL(k_ba_schur_matvec, dim3(gridP_), dim3(kBlock), 0, s, g_, R_, t_, X_, par_, ws->wrob.get(),
                         ws->hinv.get(), ws->cg_p.get(), ws->cg_y.get());
      if (timed) ctx_->prof.end(s);
      if (multi) allreduce_sum(ctx_, ws->cg_y.get(), n_);
      hipLaunchKernelGGL(k_cg_iter, dim3(1), dim3(kCgThreads), 0, s, n_, ws->cg_y.get(), ws->cg_p.get(),
                         ws->cg_x.get(), ws->cg_r.get(), ws->cg_z.get(), ws->dvec.get(), bj_, ws->cg.get(), tol * tol,
                         yscale);
      if ((it + 1) % chunk == 0 || it + 1 == max_iter) {
        GSFM_HIP_CHECK(hipMemcpyAsync(h, ws->cg.get(), sizeof(CgState), hipMemcpyDeviceToHost, s));
        GSFM_HIP_CHECK(hipStreamSynchronize(s));
        GSFM_HIP_CHECK(hipGetLastError());
        ctx_->prof.harvest();
        if (h->done) return h->iters;
      }
    }
    return max_iter;
  }

  gsfm_ctx* ctx_;
  BaWs* ws_;
  gsfm_ba_options opt_;
  BaParams g_{};
  BlockJacobi bj_{};
  int N_ = 0, K_ = 0, n_ = 0;
  long P_ = 0, M_ = 0, m_used_ = 0;
  int gridP_ = 1, gridN_ = 1;
  double *q_ = nullptr, *qn_ = nullptr, *t_ = nullptr, *tn_ = nullptr, *R_ = nullptr, *Rn_ = nullptr, *X_ = nullptr,
         *Xn_ = nullptr, *par_ = nullptr, *parn_ = nullptr;
};

int ba_solve_impl(gsfm_ctx* ctx, const gsfm_ba_problem* prob, const gsfm_ba_options* opt, double* cam_q,
                  double* cam_t, double* pt_xyz, double* intr, gsfm_report* rep) {
  GSFM_REQUIRE(prob && opt && cam_q && cam_t && pt_xyz && intr, "BA: null argument");
  if (prob->num_cams <= 0) throw StatusError(GSFM_ERR_EMPTY_PROBLEM, "BA: no images");  // ba.cc:17-20
  if (prob->num_pts <= 0 || prob->num_obs <= 0) throw StatusError(GSFM_ERR_EMPTY_PROBLEM, "BA: no tracks");  // ba.cc:21-24
  const double t0 = now_seconds();
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  BaSolver solver(ctx, *opt);
  solver.setup(prob, cam_q, cam_t, pt_xyz, intr);
  if (solver.used_observations() == 0 && ctx->comm.world == 1)
    throw StatusError(GSFM_ERR_EMPTY_PROBLEM, "BA: no track with enough views");
  const double t1 = now_seconds();
  const int rc = lm_minimize(solver, opt->lm, rep);
  solver.write_back(prob, cam_q, cam_t, pt_xyz, intr);
  const double t2 = now_seconds();
  if (rep) {
    rep->seconds_total = t2 - t0;
    rep->seconds_solve = t2 - t1;
  }
  return rc;
}

}  // namespace
}  // namespace gsfm

using namespace gsfm;

extern "C" void gsfm_ba_options_default(gsfm_ba_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  lm_options_default(&o->lm, 200);  // bundle_adjustment.h:31
  o->thres_loss_function = 1.0;     // bundle_adjustment.h:30
  o->optimize_rotations = 1;
  o->optimize_translation = 1;
  o->optimize_intrinsics = 1;
  o->optimize_principal_point = 0;
  o->optimize_points = 1;
  o->min_num_view_per_track = 3;
}

extern "C" int gsfm_ba_solve(gsfm_ctx* ctx, const gsfm_ba_problem* prob, const gsfm_ba_options* opt,
                             double* cam_q_inout, double* cam_t_inout, double* pt_xyz_inout,
                             double* intr_params_inout, gsfm_report* report) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  if (report) std::memset(report, 0, sizeof(*report));
  return guarded(ctx, report, [&] {
    return ba_solve_impl(ctx, prob, opt, cam_q_inout, cam_t_inout, pt_xyz_inout, intr_params_inout, report);
  });
}
