// ba_impl.hpp — global bundle adjustment on MI355X (gfx950): the solver behind gsfm_ba_solve, compiled as ba.hip (8-wide
// intrinsics blocks) and ba_wide.hip (16-wide) — see the note on GSFM_BA_KP below the layout tables.
//
// Replaces BundleAdjuster::Solve (glomap/estimators/bundle_adjustment.cc:11-106); trivial rigs below, calibrated rigs
// (RigReprojErrorConstantRigCostFunctor) and optimised cam_from_rig blocks (RigReprojErrorCostFunctor, optimize_rig_poses)
// in the "calibrated rigs" section further down — the sweeps never change, images play the cameras:
//   residual   colmap::ReprojErrorCostFunctor<CameraModel> (ba.cc:135-146): r = ImgFromCam(params, R(q) X + t) - obs
//   unknowns   frame pose (EigenQuaternionManifold tangent 3 + translation 3), point (3),
//              shared intrinsics blocks (principal point frozen unless optimize_principal_point)
//   constants  first frame q and t (ba.cc:261-266); all q when !optimize_rotations (gm.cc:208)
//   loss       Huber(1 px) (bundle_adjustment.h:30)
//   solver     Ceres LM + SPARSE_SCHUR  ->  lm.hpp + 3x3 point elimination + implicit-Schur PCG on the
//              reduced camera system (6 per frame + 8 per intrinsics block), block-Jacobi preconditioned.
//
// With a = R X, h = Jx^T g:  J_rot v = Jx (2 v x a),  J_rot^T g = 2 a x h,  J_trn = Jx,  J_pt = Jx R,
// J_intr = Jp.  Reduced-system vector layout: [6 per frame (rot, trn) | 8 per intrinsics block];
// constant or non-existent entries keep a zero Jacobian column (their step is exactly 0).
//
// Every reduction is atomic-free and in a fixed order (obsgraph.hpp):
//   track-major sweeps (point side)  use the Jacobians STORED once per linearisation as planes of
//     double2 (row 0, row 1) per column: 6 pose + 3 point + F free-intrinsics columns + the residual,
//     all pre-scaled by sqrt(rho') — HBM capacity (288 GB) buys a pure streaming PCG phase A;
//   camera-major sweeps (camera side) run one wave per camera with R, t, intrinsics and the CG
//     vector of that camera in registers and RECOMPUTE the Jacobian from the gathered 64-byte point
//     record — cheaper than streaming a second copy of the planes.
//
// Data layout in HBM (f64 unless noted):
//   track-major : pt_offset[P+1] i64, obs_cam[M] i32, obs_xy[M][2]            inputs
//                 jt[(10+F) planes][Mp] double2                                 per linearisation
//   camera-major: coff[N+2], c_src[M], c_pt[M] i32, c_xy[M][2]                 static per solve
//                 c_w[M] robust weights                                         per linearisation
//   per track   : X[P][3] (+ candidate), ptH[P][9] = (H_pp, g_p), ptdiag[P][3], ptjs[P][3],
//                 ptb[P][16] = (X, e, H_pp^-1, D_p, slot) build record (one 128-byte line), ptrec[P][8] = (X, t_p, pad) PCG record
//   per camera  : q[N][4], t[N][3], camR[N][9] (+ candidates), cam_intr[N] i32, yi_part[N][8]
//   per intr    : par[K][8] (+ candidate), intr_model[K] i32, intr_free[K] u8, intr_map[K][8] i8,
//                 ioff[K+1], icams[N] (cameras grouped by intrinsics block)
//   reduced     : diag, js, dvec, grad, gred, rhs [6N+8K]; spose[N][21], iacc44[K][44], minv[36N+64K];
//                 PCG vectors x, r, z, p, s, w (cg.hpp)
//
// This file is the body of TWO translation units (ba.hip, ba_wide.hip): GSFM_BA_KP, defined by the including file, is the
// width KP of an intrinsics block — 8 (GSFM_CAMERA_MAX_PARAMS: the nine models with at most eight parameters, every
// measured configuration) or 16 (GSFM_CAMERA_MAX_PARAMS_WIDE: FULL_OPENCV, THIN_PRISM_FISHEYE, RAD_TAN_THIN_PRISM_FISHEYE,
// colmap/sensor/models.h).  Everything below is in an anonymous namespace, so each unit has its own kernels, solver class
// and ba_solve_impl(); gsfm_ba_solve (ba.hip) picks the unit by gsfm_ba_problem::intr_stride.  The layouts written above
// for KP = 8 scale as: reduced vector [6 per frame | KP per block], per-camera shares 2 KP (linearise) and
// kIntrAcc = KP + KP (KP + 1) / 2 (build), block inverses KP x KP.  The 16-wide unit runs separate pose / intrinsics blocks only (no
// joint blocks: the 16-lane kernels hold 6 + 8 rows), deflates the gauge modes with A W formed by operator applications (the
// closed-form sweep is 8-wide), and has one projection instance (WIDE).
#include <algorithm>
#include <numeric>
#include <type_traits>

#include "camera.hpp"
#include "cg.hpp"
#include "dump.hpp"
#include "lm.hpp"
#include "obsgraph.hpp"
#include "ra_dense.hpp"  // dense_spd_solve: the reduced system of small chain-like problems, inverted by the block sweep

#ifndef GSFM_BA_KP
#error "ba_impl.hpp is included by ba.hip (GSFM_BA_KP 8) and ba_wide.hip (GSFM_BA_KP 16)"
#endif

namespace gsfm {
namespace {

constexpr int KP = GSFM_BA_KP;                      // doubles per intrinsics block
static_assert(KP == 8 || KP == 16, "intrinsics blocks are 8 or 16 wide");
constexpr int kIntrAcc = KP + (KP * (KP + 1)) / 2;  // reduced gradient | upper triangle of the Schur block (44 for KP = 8)
using FreeBits = std::conditional_t<KP == 8, unsigned char, unsigned short>;  // bit j set = parameter j is optimised
using ObsGeom = ObsGeomT<KP>;

struct BaDev {
  ObsGraph g;
  int K, F;
  long Mp;  // plane stride (observations, padded)
  const double* xy;
  const double* c_xy;
  const int* cam_intr;
  const int* intr_model;
  const FreeBits* intr_free;       // [K] bit j set = params[j] is optimised
  const signed char* intr_map;     // [K][8] compact column -> parameter index, -1 = unused column
  const int* ioff;                 // [K+1]
  const int* icams;                // [N]
  const int* obs_ik;               // [M] intrinsics block of each observation (track-major)
  const double* zrec;              // joint mode: [N][6 + F] gather record (z pose | z free intrinsics), else null
  int fixed_cam;
  const unsigned char* img_fixed;  // calibrated rigs: [images] 1 = image of the constant frame (fixed_cam is -1 then); else null
  int opt_rot, opt_trn, opt_pts;
  double huber_a;
  double lm_lo, lm_hi;
};

// plane indices of the stored Jacobians
constexpr int kPtbBa = 16;  // doubles per point build record: X (3) | e (3) | H_pp^-1 (6) | D_p (3) | slot in the constant camera's table (-1: not seen by it)

constexpr int PL_A = 0;  // 6 pose columns
constexpr int PL_B = 6;  // 3 point columns
constexpr int PL_I = 9;  // F free-intrinsics columns, then the residual plane

__device__ __forceinline__ int sym6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }
__device__ __forceinline__ int sym8(int i, int j) { return i * 8 - (i * (i - 1)) / 2 + (j - i); }  // (joint 14 x 14 blocks: KP = 8 only)
__device__ __forceinline__ int symK(int i, int j) { return i * KP - (i * (i - 1)) / 2 + (j - i); }

struct ObsJac {
  double Jpose[2][6];  // [rot | trn], masked
  double Jpt[2][3];    // masked by opt_pts
};

__device__ __forceinline__ void build_jac(const BaDev& g, int n, const double* __restrict__ R9, const ObsGeom& o,
                                          ObsJac& J) {
  const bool fixed = g.img_fixed ? g.img_fixed[n] != 0 : n == g.fixed_cam;
  const bool rf = g.opt_rot && !fixed;
  const bool tf = g.opt_trn && !fixed;
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

__device__ __forceinline__ void mask_intr(FreeBits bits, double (&Jp)[2][KP]) {
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    if (!((bits >> j) & 1)) {
      Jp[0][j] = 0.0;
      Jp[1][j] = 0.0;
    }
  }
}

// v[idx] with a register-resident array and a run-time index (select chain, no scratch)
__device__ __forceinline__ double sel8(const double (&v)[KP], int idx) {
  double r = 0.0;
#pragma unroll
  for (int p = 0; p < KP; ++p) r = (p == idx) ? v[p] : r;
  return r;
}

struct Map8 {
  signed char m[KP];
};
__device__ __forceinline__ Map8 load_map(const signed char* __restrict__ p) {
  Map8 r;
#pragma unroll
  for (int h = 0; h < KP / 8; ++h) {
    const unsigned long long bits = reinterpret_cast<const unsigned long long*>(p)[h];
#pragma unroll
    for (int j = 0; j < 8; ++j) r.m[8 * h + j] = (signed char)((bits >> (8 * j)) & 0xff);
  }
  return r;
}

__device__ __forceinline__ double lm_damping(double h, double js, double radius, double lo, double hi) {
  const double j2 = js * js;
  return fmin(fmax(j2 * h, lo), hi) / (radius * j2);
}

__device__ __forceinline__ double block_max(double v, double* smem /* >= 4 */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(smem[0], smem[1]), fmax(smem[2], smem[3]));
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

// ---- linearize, point side ----------------------------------------------------------------------
// One lane per observation (track-major tiles): cost, the stored Jacobian planes (scaled by sqrt(w)),
// and H_pp / g_p per track through a segmented wave scan.  All plane writes are coalesced.
// part[block][2] = {cost, max |g_pt|}.
template <int F, bool WIDE>
__global__ void __launch_bounds__(kBlock)
    k_ba_lin_track(BaDev g, const double* __restrict__ camR, const double* __restrict__ t,
                   const double* __restrict__ X, const double* __restrict__ par, double2* __restrict__ jt,
                   double* __restrict__ ptdiag, double* __restrict__ ptH, double* __restrict__ part) {
  __shared__ double smem[8];
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  double cost = 0.0, gmax = 0.0;
  for (int tile = wave; tile < g.g.T; tile += nwaves) {
    const long k0 = g.g.tile_k[tile], k1 = g.g.tile_k[tile + 1];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // H (xx xy xz yy yz zz) | g
    int key = -1 - lane;
    for (long k = k0 + lane; k < k1; k += 64) {
      const int p = g.g.obs_pt[k];
      key = p;
      if (!g.g.used[p]) continue;
      const int n = g.g.cam[k];
      const int ik = g.cam_intr[n];
      const double* R9 = camR + 9 * (long)n;
      ObsGeom o;
      obs_geom<WIDE, KP>(R9, t + 3 * (long)n, ld3(X + 3 * (long)p), g.intr_model[ik], par + KP * (long)ik, o);
      const double2 ob = *reinterpret_cast<const double2*>(g.xy + 2 * k);
      const double r0 = o.valid ? o.px - ob.x : 0.0;
      const double r1 = o.valid ? o.py - ob.y : 0.0;
      double rho, w;
      huber(g.huber_a, 1.0, r0 * r0 + r1 * r1, rho, w);
      if (!o.valid) w = 0.0;
      cost += 0.5 * rho;
      ObsJac J;
      build_jac(g, n, R9, o, J);
      const double sw = sqrt(w);
#pragma unroll
      for (int j = 0; j < 6; ++j) jt[(PL_A + j) * g.Mp + k] = make_double2(sw * J.Jpose[0][j], sw * J.Jpose[1][j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) jt[(PL_B + j) * g.Mp + k] = make_double2(sw * J.Jpt[0][j], sw * J.Jpt[1][j]);
      if constexpr (F > 0) {
        const Map8 mp = load_map(g.intr_map + KP * (long)ik);
#pragma unroll
        for (int j = 0; j < F; ++j) {
          const int pm = mp.m[j];
          jt[(PL_I + j) * g.Mp + k] = make_double2(sw * sel8(o.Jp[0], pm), sw * sel8(o.Jp[1], pm));
        }
      }
      jt[(PL_I + F) * g.Mp + k] = make_double2(sw * r0, sw * r1);
      acc[0] += w * (J.Jpt[0][0] * J.Jpt[0][0] + J.Jpt[1][0] * J.Jpt[1][0]);
      acc[1] += w * (J.Jpt[0][0] * J.Jpt[0][1] + J.Jpt[1][0] * J.Jpt[1][1]);
      acc[2] += w * (J.Jpt[0][0] * J.Jpt[0][2] + J.Jpt[1][0] * J.Jpt[1][2]);
      acc[3] += w * (J.Jpt[0][1] * J.Jpt[0][1] + J.Jpt[1][1] * J.Jpt[1][1]);
      acc[4] += w * (J.Jpt[0][1] * J.Jpt[0][2] + J.Jpt[1][1] * J.Jpt[1][2]);
      acc[5] += w * (J.Jpt[0][2] * J.Jpt[0][2] + J.Jpt[1][2] * J.Jpt[1][2]);
      acc[6] += w * (J.Jpt[0][0] * r0 + J.Jpt[1][0] * r1);
      acc[7] += w * (J.Jpt[0][1] * r0 + J.Jpt[1][1] * r1);
      acc[8] += w * (J.Jpt[0][2] * r0 + J.Jpt[1][2] * r1);
    }
    seg_scan<9>(acc, key, lane);
    if (seg_is_tail(key, lane) && key >= 0 && g.g.used[key]) {
      const long p = key;
      ptdiag[3 * p] = acc[0];
      ptdiag[3 * p + 1] = acc[3];
      ptdiag[3 * p + 2] = acc[5];
      double* hp = ptH + 9 * p;
#pragma unroll
      for (int j = 0; j < 9; ++j) hp[j] = acc[j];
      gmax = fmax(gmax, fmax(fabs(acc[6]), fmax(fabs(acc[7]), fabs(acc[8]))));
    }
  }
  double v[1] = {cost};
  block_sum<1>(v, smem);
  const double m = block_max(gmax, smem + 4);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = v[0];
    part[blockIdx.x * 2 + 1] = m;
  }
}

// ---- linearize, camera side -------------------------------------------------------------------------
// One wave per camera: robust weights in camera-major order (c_w), squared column norms and gradient
// of the pose block; the intrinsics share of this camera goes to ipart[n][2 KP] = (diag KP | grad KP).
// GRAM (optimised cam_from_rig blocks): the pose part is the full 6 x 6 Gram matrix sum w J^T J of the IMAGE's own
// tangent (upper triangle, sym6 order) in diag[n][21] and its gradient in grad[n][6]; the frame / sensor column norms
// and gradients are congruences of it (k_ba_rig_frame_lin, k_ba_rig_sensor_lin).
template <bool GRAM, bool WIDE>
__global__ void __launch_bounds__(kBlock)
    k_ba_lin_cam(BaDev g, const double* __restrict__ camR, const double* __restrict__ t,
                 const double* __restrict__ X, const double* __restrict__ par, double* __restrict__ c_w,
                 double* __restrict__ diag, double* __restrict__ grad, double* __restrict__ ipart,
                 const double* __restrict__ sensR /* calibrated rigs: [images][12] cam_from_rig (R row-major | t), else null */) {
  constexpr int PW = GRAM ? 27 : 12;  // pose accumulators
  constexpr int W = PW + 2 * KP;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  for (int it = wave; it < cam_seg_count(g.g); it += nwaves) {
    const int sg = cam_seg_index(g.g, it);
    const int n = g.g.seg_cam[sg];
    const int ik = g.cam_intr[n];
    const int model = g.intr_model[ik];
    const FreeBits bits = g.intr_free[ik];
    const double* R9 = camR + 9 * (long)n;
    const double* t3 = t + 3 * (long)n;
    const double* pp = par + KP * (long)ik;
    double acc[W];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] = 0.0;
    for (int k = cam_seg_k0(g.g, sg) + lane; k < cam_seg_k1(g.g, sg); k += 64) {
      ObsGeom o;
      obs_geom<WIDE, KP>(R9, t3, ld3(X + 3 * (long)g.g.c_pt[k]), model, pp, o);
      const double r0 = o.valid ? o.px - g.c_xy[2 * (long)k] : 0.0;
      const double r1 = o.valid ? o.py - g.c_xy[2 * (long)k + 1] : 0.0;
      double rho, w;
      huber(g.huber_a, 1.0, r0 * r0 + r1 * r1, rho, w);
      if (!o.valid) w = 0.0;
      c_w[k] = w;
      ObsJac J;
      build_jac(g, n, R9, o, J);
      if (!GRAM && sensR != nullptr) {
        // the pose unknown is the FRAME's: its tangent d_f maps to this image's tangent as (R_s d_rot, R_s d_trn), so the
        // columns of the frame Jacobian are J_image R_s — squared norms and gradient are accumulated in that basis
        const double* Rs = sensR + 12 * (long)n;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const double a0 = J.Jpose[r][3 * h], a1 = J.Jpose[r][3 * h + 1], a2 = J.Jpose[r][3 * h + 2];
#pragma unroll
            for (int j = 0; j < 3; ++j) J.Jpose[r][3 * h + j] = a0 * Rs[j] + a1 * Rs[3 + j] + a2 * Rs[6 + j];
          }
        }
      }
      const double g0 = w * r0, g1 = w * r1;
      if constexpr (GRAM) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
          for (int j = i; j < 6; ++j)
            acc[i * 6 - (i * (i - 1)) / 2 + (j - i)] += w * (J.Jpose[0][i] * J.Jpose[0][j] + J.Jpose[1][i] * J.Jpose[1][j]);
          acc[21 + i] += J.Jpose[0][i] * g0 + J.Jpose[1][i] * g1;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          acc[j] += w * (J.Jpose[0][j] * J.Jpose[0][j] + J.Jpose[1][j] * J.Jpose[1][j]);
          acc[6 + j] += J.Jpose[0][j] * g0 + J.Jpose[1][j] * g1;
        }
      }
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        if ((bits >> j) & 1) {
          acc[PW + j] += w * (o.Jp[0][j] * o.Jp[0][j] + o.Jp[1][j] * o.Jp[1][j]);
          acc[PW + KP + j] += o.Jp[0][j] * g0 + o.Jp[1][j] * g1;
        }
      }
    }
    wave_allsum<W>(acc);
    if (!cam_seg_total<W>(g.g, sg, acc, lane)) continue;
    if (lane == 0) {
      if constexpr (GRAM) {
#pragma unroll
        for (int j = 0; j < 21; ++j) diag[21 * (long)n + j] = acc[j];
#pragma unroll
        for (int j = 0; j < 6; ++j) grad[6 * (long)n + j] = acc[21 + j];
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          diag[6 * (long)n + j] = acc[j];
          grad[6 * (long)n + j] = acc[6 + j];
        }
      }
#pragma unroll
      for (int j = 0; j < 2 * KP; ++j) ipart[2 * KP * (long)n + j] = acc[PW + j];
    }
  }
}

// out[k][0..W) = sum over the cameras of intrinsics block k of part[cam][0..W)   (one block per group,
// fixed order: thread-strided partial sums, then the block tree)
template <int W>
__global__ void __launch_bounds__(kBlock)
    k_ba_group_sum(int K, const int* __restrict__ ioff, const int* __restrict__ icams,
                   const double* __restrict__ part, double* __restrict__ out) {
  if constexpr (W <= 64) {
  __shared__ double smem[4 * W];
  for (int k = blockIdx.x; k < K; k += gridDim.x) {
    double acc[W];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] = 0.0;
    for (int i = ioff[k] + threadIdx.x; i < ioff[k + 1]; i += blockDim.x) {
      const double* s = part + (long)W * icams[i];
#pragma unroll
      for (int j = 0; j < W; ++j) acc[j] += s[j];
    }
    block_sum<W>(acc, smem);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < W; ++j) out[(long)W * k + j] = acc[j];
    }
    __syncthreads();
  }
  } else {
    // the 16-wide unit's rows (152, 112 values): eight columns at a time — every column is summed in the same order as
    // above, the fully unrolled block tree over all columns was 4e5 instructions
    constexpr int CH = 8;
    static_assert(W % CH == 0, "row width must be a multiple of the column chunk");
    __shared__ double smem[4 * CH];
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
      for (int c0 = 0; c0 < W; c0 += CH) {
        double acc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = 0.0;
        for (int i = ioff[k] + threadIdx.x; i < ioff[k + 1]; i += blockDim.x) {
          const double* s = part + (long)W * icams[i] + c0;
#pragma unroll
          for (int j = 0; j < CH; ++j) acc[j] += s[j];
        }
        block_sum<CH>(acc, smem);
        if (threadIdx.x == 0) {
#pragma unroll
          for (int j = 0; j < CH; ++j) out[(long)W * k + c0 + j] = acc[j];
        }
        __syncthreads();
      }
    }
  }
}

// Same sum with one THREAD per group, for problems whose groups are small (e.g. one intrinsics
// block per image): sequential fixed-order sum over the group's cameras.
template <int W>
__global__ void __launch_bounds__(kBlock)
    k_ba_group_sum_small(int K, const int* __restrict__ ioff, const int* __restrict__ icams,
                         const double* __restrict__ part, double* __restrict__ out) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
    double acc[W];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] = 0.0;
    for (int i = ioff[k]; i < ioff[k + 1]; ++i) {
      const double* s = part + (long)W * icams[i];
#pragma unroll
      for (int j = 0; j < W; ++j) acc[j] += s[j];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) out[(long)W * k + j] = acc[j];
  }
}

static __global__ void __launch_bounds__(kBlock)
    k_ba_obs_ik(long M, const int* __restrict__ cam, const int* __restrict__ cam_intr, int* __restrict__ obs_ik) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < M; k += (long)gridDim.x * blockDim.x)
    obs_ik[k] = cam_intr[cam[k]];
}

// scatter iacc16 [K][16] into the reduced-vector layout: diag/grad [6N + 8k + j]
__global__ void __launch_bounds__(kBlock)
    k_ba_intr_unpack16(int N, int K, const double* __restrict__ intr_acc, double* __restrict__ diag,
                       double* __restrict__ grad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < KP * K; i += gridDim.x * blockDim.x) {
    const int k = i / KP, j = i % KP;
    diag[6 * (long)N + i] = intr_acc[2 * KP * (long)k + j];
    grad[6 * (long)N + i] = intr_acc[2 * KP * (long)k + KP + j];
  }
}

// max_i |vec[i]| per block -> mpart[block]  (the gradient has 10^4 .. 10^6 entries: one workgroup took 140 us)
__global__ void __launch_bounds__(kBlock) k_ba_absmax(const double* __restrict__ vec, int nvec, double* __restrict__ mpart) {
  __shared__ double smem[4];
  double m = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) m = fmax(m, fabs(vec[i]));
  m = block_max(m, smem);
  if (threadIdx.x == 0) mpart[blockIdx.x] = m;
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
  const double m = block_max(gmax, smem + 4);
  if (threadIdx.x == 0) {
    out[0] = v[0];
    out[1] = m;
  }
}

// js = 1 / (1 + sqrt(diag)) (Ceres jacobi_scaling, fixed at the initial point)
__global__ void __launch_bounds__(kBlock)
    k_ba_jacobi(long n, int enabled, const double* __restrict__ diag, double* __restrict__ js) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    js[i] = enabled ? 1.0 / (1.0 + sqrt(diag[i])) : 1.0;
}

// ---- build, point side (radius dependent) -----------------------------------------------------------
// One thread per track: H_pp + damping -> inverse, e = H_pp^-1 g_p; writes the two point records.
__global__ void __launch_bounds__(kBlock)
    k_ba_build_track(BaDev g, double radius, const double* __restrict__ X, const double* __restrict__ ptH,
                     const double* __restrict__ ptdiag, const double* __restrict__ ptjs,
                     double* __restrict__ ptb, double* __restrict__ ptrec, double* __restrict__ pth) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < g.g.P; p += (long)gridDim.x * blockDim.x) {
    if (!g.g.used[p]) continue;
    const V3 Xp = ld3(X + 3 * p);
    const double* hp = ptH + 9 * p;
    S3 H{hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
    S3 Hi{0, 0, 0, 0, 0, 0};
    V3 e{0, 0, 0};
    V3 D{0, 0, 0};
    if (g.opt_pts) {
      D.x = lm_damping(ptdiag[3 * p], ptjs[3 * p], radius, g.lm_lo, g.lm_hi);
      D.y = lm_damping(ptdiag[3 * p + 1], ptjs[3 * p + 1], radius, g.lm_lo, g.lm_hi);
      D.z = lm_damping(ptdiag[3 * p + 2], ptjs[3 * p + 2], radius, g.lm_lo, g.lm_hi);
      H.xx += D.x;
      H.yy += D.y;
      H.zz += D.z;
      Hi = inv3(H);
      e = mul(Hi, ld3(hp + 6));
    }
    double* b = ptb + kPtbBa * p;
    st3(b, Xp);
    st3(b + 3, e);
    b[6] = Hi.xx; b[7] = Hi.xy; b[8] = Hi.xz; b[9] = Hi.yy; b[10] = Hi.yz; b[11] = Hi.zz;
    st3(b + 12, D);
    b[15] = -1.0;  // k_ba_fixed_share marks the points the constant camera sees
    double* hc = pth + 6 * p;  // compact copy for phase A: consecutive tracks -> one coalesced 48-byte stream
    hc[0] = Hi.xx; hc[1] = Hi.xy; hc[2] = Hi.xz; hc[3] = Hi.yy; hc[4] = Hi.yz; hc[5] = Hi.zz;
    double* pr = ptrec + 8 * p;
    st3(pr, Xp);
    pr[3] = pr[4] = pr[5] = 0.0;
  }
}

// ---- build, camera side ------------------------------------------------------------------------------
// One wave per camera: reduced gradient J_a^T w (r - J_pt e) and the diagonal Schur blocks
// J_a^T W_k J_a, W_k = w (I - w J_pt H_pp^-1 J_pt^T), for the pose block (6 + 21) and this camera's
// share of its intrinsics block (ipart[n][kIntrAcc] = gred KP | S KP (KP + 1) / 2; 44 for KP = 8).
// The intrinsics part is accumulated over the F COMPACT columns (the free parameters of the block, intr_map) and
// scattered to the 8-wide layout at the end: with SIMPLE_RADIAL and a fixed principal point (F = 2) that is 44
// accumulators per lane instead of 119 — the full-width version sat at 252 registers, one wave per SIMD, 600 - 800 us
// per launch on configs[3].
template <int F>
__device__ __forceinline__ int symF(int i, int j) { return i * F - (i * (i - 1)) / 2 + (j - i); }
template <bool JOINT, bool WIDE, int F>
__global__ void __launch_bounds__(kBlock)
    k_ba_build_cam(BaDev g, const double* __restrict__ camR, const double* __restrict__ t,
                   const double* __restrict__ par, const double* __restrict__ c_w,
                   const double* __restrict__ ptb, double* __restrict__ gred, double* __restrict__ spose,
                   double* __restrict__ ipart, double* __restrict__ scross /* [N][48], JOINT only */) {
  constexpr int NI = F + (F * (F + 1)) / 2;       // compact intrinsics gradient | upper triangle
  constexpr int OI = 27, OS = 27 + F, OC = 27 + NI;
  constexpr int NACC = JOINT ? OC + 6 * F : (OC > 27 ? OC : 27);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  for (int it = wave; it < cam_seg_count(g.g); it += nwaves) {
    const int sg = cam_seg_index(g.g, it);
    const int n = g.g.seg_cam[sg];
    const int ik = g.cam_intr[n];
    const int model = g.intr_model[ik];
    const double* R9 = camR + 9 * (long)n;
    const double* t3 = t + 3 * (long)n;
    const double* pp = par + KP * (long)ik;
    Map8 mp = load_map(g.intr_map + KP * (long)ik);
    const FreeBits bits = g.intr_free[ik];
    if constexpr (F == KP) {  // full width: the columns ARE the parameters, fixed ones masked (no select chains)
#pragma unroll
      for (int j = 0; j < KP; ++j) mp.m[j] = ((bits >> j) & 1) ? (signed char)j : (signed char)-1;
    }
    double acc[NACC];  // gred 6 | spose 21 | igred F | sii F (F + 1) / 2 | (JOINT) pose x intrinsics cross block 6 x F
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.0;
    for (int k = cam_seg_k0(g.g, sg) + lane; k < cam_seg_k1(g.g, sg); k += 64) {
      const double w = c_w[k];
      if (w == 0.0) continue;
      const double* b = ptb + kPtbBa * (long)g.g.c_pt[k];
      const V3 e = ld3(b + 3);
      const S3 Hi{b[6], b[7], b[8], b[9], b[10], b[11]};
      ObsGeom o;
      obs_geom<WIDE, KP>(R9, t3, ld3(b), model, pp, o);
      const double r0 = o.px - g.c_xy[2 * (long)k], r1 = o.py - g.c_xy[2 * (long)k + 1];
      ObsJac J;
      build_jac(g, n, R9, o, J);
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
      double Jc[2][F > 0 ? F : 1];  // the free columns of d(projection) / d(intrinsics); unused columns (-1) are zero
      if constexpr (F == KP) {
        mask_intr(bits, o.Jp);
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          Jc[0][j] = o.Jp[0][j];
          Jc[1][j] = o.Jp[1][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < F; ++j) {
          const int pm = mp.m[j];
          Jc[0][j] = sel8(o.Jp[0], pm);
          Jc[1][j] = sel8(o.Jp[1], pm);
        }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        acc[i] += w * (J.Jpose[0][i] * q0 + J.Jpose[1][i] * q1);
        const double a0 = W00 * J.Jpose[0][i] + W01 * J.Jpose[1][i];
        const double a1 = W01 * J.Jpose[0][i] + W11 * J.Jpose[1][i];
#pragma unroll
        for (int j = i; j < 6; ++j) acc[6 + sym6(i, j)] += a0 * J.Jpose[0][j] + a1 * J.Jpose[1][j];
        if constexpr (JOINT) {
#pragma unroll
          for (int j = 0; j < F; ++j) acc[OC + F * i + j] += a0 * Jc[0][j] + a1 * Jc[1][j];
        }
      }
#pragma unroll
      for (int i = 0; i < F; ++i) {
        acc[OI + i] += w * (Jc[0][i] * q0 + Jc[1][i] * q1);
        const double a0 = W00 * Jc[0][i] + W01 * Jc[1][i];
        const double a1 = W01 * Jc[0][i] + W11 * Jc[1][i];
#pragma unroll
        for (int j = i; j < F; ++j) acc[OS + symF<F>(i, j)] += a0 * Jc[0][j] + a1 * Jc[1][j];
      }
    }
    wave_allsum<NACC>(acc);
    if (!cam_seg_total<NACC>(g.g, sg, acc, lane)) continue;
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j) gred[6 * (long)n + j] = acc[j];
#pragma unroll
      for (int j = 0; j < 21; ++j) spose[21 * (long)n + j] = acc[6 + j];
      // compact -> 8-wide (intr_map is increasing: column i < column j means parameter index p_i < p_j)
      double* ip = ipart + kIntrAcc * (long)n;
#pragma unroll
      for (int j = 0; j < kIntrAcc; ++j) ip[j] = 0.0;
      if constexpr (JOINT) {
#pragma unroll
        for (int j = 0; j < 48; ++j) scross[48 * (long)n + j] = 0.0;
      }
#pragma unroll
      for (int i = 0; i < F; ++i) {
        const int pi = mp.m[i];
        if (pi < 0) continue;
        ip[pi] = acc[OI + i];
#pragma unroll
        for (int j = i; j < F; ++j) {
          const int pj = mp.m[j];
          if (pj >= 0) ip[KP + symK(pi, pj)] = acc[OS + symF<F>(i, j)];
        }
        if constexpr (JOINT) {
#pragma unroll
          for (int a = 0; a < 6; ++a) scross[48 * (long)n + 8 * a + pi] = acc[OC + F * a + i];
        }
      }
    }
  }
}

// One thread per block of the block-Jacobi preconditioner: damping, rhs = -g', dense inverse.
__global__ void __launch_bounds__(kBlock)
    k_ba_blocks_finalize(int N, int K, double radius, double lo, double hi, const double* __restrict__ diag,
                         const double* __restrict__ js, const double* __restrict__ gred,
                         const double* __restrict__ spose, const double* __restrict__ intr_acc,
                         double* __restrict__ dvec, double* __restrict__ rhs, double* __restrict__ minv) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < N + K; b += gridDim.x * blockDim.x) {
    double A[KP * KP];
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
      spd_inverse<KP>(A, 6);
      double* m = minv + 36 * (long)b;
      for (int i = 0; i < 36; ++i) m[i] = A[i];
    } else {
      const int k = b - N;
      const long o = 6 * (long)N + KP * (long)k;
      const double* acc = intr_acc + kIntrAcc * (long)k;
      for (int i = 0; i < KP; ++i) {
        const double D = lm_damping(diag[o + i], js[o + i], radius, lo, hi);
        dvec[o + i] = D;
        rhs[o + i] = -acc[i];
        for (int j = i; j < KP; ++j) {
          const double v = acc[KP + symK(i, j)] + (i == j ? D : 0.0);
          A[i * KP + j] = v;
          A[j * KP + i] = v;
        }
      }
      spd_inverse<KP>(A, KP);
      double* m = minv + 36 * (long)N + KP * KP * (long)k;
      for (int i = 0; i < KP * KP; ++i) m[i] = A[i];
    }
  }
}

// Joint variant (one intrinsics block per camera): damping and rhs as above, and the inverse of the
// 14 x 14 block [S_pose + D, C; C^T, S_intr + D] of camera n and intrinsics block cam_intr[n], stored
// transposed (minvj[(i * 14 + j) * N + n]) for coalesced reads in k_cg_update_joint.
// 16 lanes per camera: lane i < 14 keeps row i in registers, Gauss-Jordan (SPD: no pivoting) with the
// pivot row broadcast by wave shuffles — no scratch, 4 cameras per wave.
__global__ void __launch_bounds__(kBlock)
    k_ba_blocks_finalize_joint(int N, double radius, double lo, double hi, const int* __restrict__ cam_intr,
                               const double* __restrict__ diag, const double* __restrict__ js,
                               const double* __restrict__ gred, const double* __restrict__ spose,
                               const double* __restrict__ intr_acc, const double* __restrict__ scross,
                               double* __restrict__ dvec, double* __restrict__ rhs, double* __restrict__ minvj) {
  const int lane = threadIdx.x & 63;
  const int base = lane & ~15;
  const int c = threadIdx.x >> 4, i = threadIdx.x & 15;
  for (int n0 = blockIdx.x * (kBlock / 16); n0 < N; n0 += gridDim.x * (kBlock / 16)) {
    const int n = n0 + c;
    const bool cam_ok = n < N;
    const bool act = cam_ok && i < 14;
    double a[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) a[j] = (j == i) ? 1.0 : 0.0;  // idle lanes / cameras: identity rows
    if (act) {
      const int k = cam_intr[n];
      const long op = 6 * (long)n, oi = 6 * (long)N + 8 * (long)k;
      const double* sp = spose + 21 * (long)n;
      const double* acc = intr_acc + 44 * (long)k;
      const double* cr = scross + 48 * (long)n;
      if (i < 6) {
        const double D = lm_damping(diag[op + i], js[op + i], radius, lo, hi);
        dvec[op + i] = D;
        rhs[op + i] = -gred[op + i];
#pragma unroll
        for (int j = 0; j < 6; ++j) a[j] = sp[sym6(i < j ? i : j, i < j ? j : i)] + (i == j ? D : 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[6 + j] = cr[8 * i + j];
      } else {
        const int ii = i - 6;
        const double D = lm_damping(diag[oi + ii], js[oi + ii], radius, lo, hi);
        dvec[oi + ii] = D;
        rhs[oi + ii] = -acc[ii];
#pragma unroll
        for (int j = 0; j < 6; ++j) a[j] = cr[8 * j + ii];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[6 + j] = acc[8 + sym8(ii < j ? ii : j, ii < j ? j : ii)] + (ii == j ? D : 0.0);
      }
    }
    double d0 = 0.0;
#pragma unroll
    for (int j = 0; j < 14; ++j) d0 = (j == i) ? a[j] : d0;  // own diagonal entry (fallback)
    bool bad = false;
#pragma unroll
    for (int p = 0; p < 14; ++p) {
      double rowp[14];
#pragma unroll
      for (int j = 0; j < 14; ++j) rowp[j] = __shfl(a[j], base + p, 64);
      const double piv = rowp[p];
      bad = bad || !(piv > 0.0) || !isfinite(piv);
      const double ipiv = 1.0 / piv;
      const double colv = a[p];
      const bool is_p = i == p;
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        double v = is_p ? rowp[j] * ipiv : a[j] - colv * rowp[j] * ipiv;
        if (j == p) v = is_p ? ipiv : -colv * ipiv;
        a[j] = v;
      }
    }
    if (act) {
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const double v = bad ? ((j == i) ? 1.0 / d0 : 0.0) : a[j];  // indefinite block: fall back to its diagonal
        minvj[(size_t)(i * 14 + j) * N + n] = v;
      }
    }
  }
}

// ---- the hot pair: w = (H_aa - H_ap H_pp^-1 H_pa + D) z ---------------------------------------------
// Phase A, track-major, one lane per observation over the stored planes:
//   u_k = A_k z_{c(k)} + I_k z_{intr(k)},   t_p = H_pp^-1 sum_k B_k^T u_k  ->  ptrec[p].t
// Algorithmic bytes per observation: (9 + F) double2 planes = 16 (9 + F), + cam 4 + pt 4; per track 24
// written; the gathers of z (48 B per camera, 8 F per intrinsics block) are L2-resident.
// 16-byte plane load; NT = non-temporal (the planes are read once per sweep and never fit a cache)
typedef double ba_d2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 ld_plane(const double2* __restrict__ p) {
  if constexpr (NT) {
    const ba_d2v v = __builtin_nontemporal_load(reinterpret_cast<const ba_d2v*>(p));
    return make_double2(v.x, v.y);
  } else {
    return *p;
  }
}

template <int F, bool NT>
__global__ void __launch_bounds__(kBlock)
    k_ba_phaseA(BaDev g, CgVec v, int it, double tol2, const double2* __restrict__ jt,
                const double* __restrict__ pth, double* __restrict__ ptrec) {
  __shared__ double smem[4 * 2 + 2];
  if (cg_converged(v, it, tol2, smem)) return;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  const double* zintr = v.z + 6 * (long)g.g.N;
  for (int tile = wave; tile < g.g.T; tile += nwaves) {
    const long k0 = g.g.tile_k[tile], k1 = g.g.tile_k[tile + 1];
    double acc[3] = {0, 0, 0};
    int key = -1 - lane;
    for (long k = k0 + lane; k < k1; k += 64) {
      // all (9 + F) plane loads are issued before the dependent camera gathers: the streams are what bounds the
      // sweep, so they must be in flight while the index -> z-record chain resolves
      double2 pa[6], pi[F > 0 ? F : 1], pb[3];
#pragma unroll
      for (int j = 0; j < 6; ++j) pa[j] = ld_plane<NT>(jt + (PL_A + j) * g.Mp + k);
#pragma unroll
      for (int j = 0; j < F; ++j) pi[j] = ld_plane<NT>(jt + (PL_I + j) * g.Mp + k);
#pragma unroll
      for (int j = 0; j < 3; ++j) pb[j] = ld_plane<NT>(jt + (PL_B + j) * g.Mp + k);
      key = g.g.obs_pt[k];
      const long n = g.g.cam[k];
      double u0 = 0.0, u1 = 0.0;
      if (g.zrec != nullptr) {
        // joint mode: z of the pose and of the free intrinsics of camera n in ONE (6 + F)-double record
        const double2* zp = reinterpret_cast<const double2*>(g.zrec + (6 + F) * n);
        double zz[6 + F];
#pragma unroll
        for (int j = 0; j < (6 + F) / 2; ++j) {
          const double2 t2 = zp[j];
          zz[2 * j] = t2.x;
          zz[2 * j + 1] = t2.y;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          u0 += pa[j].x * zz[j];
          u1 += pa[j].y * zz[j];
        }
#pragma unroll
        for (int j = 0; j < F; ++j) {
          u0 += pi[j].x * zz[6 + j];
          u1 += pi[j].y * zz[6 + j];
        }
      } else {
        const double2* zp = reinterpret_cast<const double2*>(v.z + 6 * n);
        const double2 z01 = zp[0], z23 = zp[1], z45 = zp[2];
        const double zz[6] = {z01.x, z01.y, z23.x, z23.y, z45.x, z45.y};
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          u0 += pa[j].x * zz[j];
          u1 += pa[j].y * zz[j];
        }
        if constexpr (F > 0) {
          const int ik = g.obs_ik[k];
          const Map8 mp = load_map(g.intr_map + KP * (long)ik);
          const double* zi = zintr + KP * (long)ik;
#pragma unroll
          for (int j = 0; j < F; ++j) {
            const int pm = mp.m[j];
            const double zv = pm >= 0 ? zi[pm] : 0.0;
            u0 += pi[j].x * zv;
            u1 += pi[j].y * zv;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] += pb[j].x * u0 + pb[j].y * u1;
    }
    seg_scan<3>(acc, key, lane);
    if (seg_is_tail(key, lane) && key >= 0 && g.g.used[key]) {
      const double* b = pth + 6 * (long)key;
      const V3 tp = mul(S3{b[0], b[1], b[2], b[3], b[4], b[5]}, V3{acc[0], acc[1], acc[2]});
      st3(ptrec + 8 * (long)key + 3, tp);
    }
  }
}

// Phase B, camera-major, one wave per camera, Jacobian recomputed from the gathered point record:
//   w_n = sum_k J_a^T w_k (J_a z_a - J_pt t_p) + D_n z_n,   yi_part[n] = this camera's intrinsics rows.
// Algorithmic bytes per observation: c_w 8 + c_pt 4 + the 64-byte point record (X_p, t_p).
template <bool WIDE>
__global__ void __launch_bounds__(kBlock)
    k_ba_phaseB(BaDev g, CgVec v, double yscale, const double* __restrict__ camR, const double* __restrict__ t,
                const double* __restrict__ par, const double* __restrict__ c_w,
                const double* __restrict__ ptrec, const double* __restrict__ dvec, double* __restrict__ yi_part,
                int dslot0 /* first delta slot of this launch: 0, or behind the other slots for the combine pass */) {
  __shared__ double sdelta[kBlock / 64];
  if (v.st->done) return;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wave = blockIdx.x * (kBlock / 64) + wid;
  const int nwaves = gridDim.x * (kBlock / 64);
  double delta = 0.0;
  for (int it = wave; it < cam_seg_count(g.g); it += nwaves) {
    const int sg = cam_seg_index(g.g, it);
    const int n = g.g.seg_cam[sg];
    const int ik = g.cam_intr[n];
    const int model = g.intr_model[ik];
    const FreeBits bits = g.intr_free[ik];
    const double* R9 = camR + 9 * (long)n;
    const double* t3 = t + 3 * (long)n;
    const double* pp = par + KP * (long)ik;
    const bool fixed = g.img_fixed ? g.img_fixed[n] != 0 : n == g.fixed_cam;
    const bool rf = g.opt_rot && !fixed, tf = g.opt_trn && !fixed;
    const double* zp = v.z + 6 * (long)n;
    const V3 zr{zp[0], zp[1], zp[2]}, zt{zp[3], zp[4], zp[5]};
    double zi[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) zi[j] = ((bits >> j) & 1) ? v.z[6 * (long)g.g.N + KP * (long)ik + j] : 0.0;
    double acc[6 + KP];
#pragma unroll
    for (int j = 0; j < 6 + KP; ++j) acc[j] = 0.0;
    // software-pipelined: the index -> 64-byte record gather of observation k + 64 is in flight while observation k
    // goes through its ~300 flops
    const int kend = cam_seg_k1(g.g, sg);
    int k = cam_seg_k0(g.g, sg) + lane;
    double w_nx = 0.0;
    V3 X_nx{0, 0, 0}, t_nx{0, 0, 0};
    if (k < kend) {
      w_nx = c_w[k];
      ld6(ptrec + 8 * (long)g.g.c_pt[k], X_nx, t_nx);  // 64-byte aligned record, three 16-byte gathers
    }
    for (; k < kend; k += 64) {
      const double w = w_nx;
      const V3 Xp = X_nx, tp = t_nx;
      if (k + 64 < kend) {
        w_nx = c_w[k + 64];
        ld6(ptrec + 8 * (long)g.g.c_pt[k + 64], X_nx, t_nx);
      }
      if (w == 0.0) continue;
      ObsGeom o;
      obs_geom<WIDE, KP>(R9, t3, Xp, model, pp, o);
      V3 om = V3{0, 0, 0} - R_mul(R9, tp);  // - J_pt t_p = - Jx (R t_p)
      if (rf) om = om + 2.0 * cross(zr, o.a);
      if (tf) om = om + zt;
      double u0, u1;
      jx_mul(o.Jx, om, u0, u1);
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        if ((bits >> j) & 1) {
          u0 += o.Jp[0][j] * zi[j];
          u1 += o.Jp[1][j] * zi[j];
        }
      }
      const double g0 = w * u0, g1 = w * u1;
      const V3 h = jxT_mul(o.Jx, g0, g1);
      if (rf) {
        const V3 yr = 2.0 * cross(o.a, h);
        acc[0] += yr.x;
        acc[1] += yr.y;
        acc[2] += yr.z;
      }
      if (tf) {
        acc[3] += h.x;
        acc[4] += h.y;
        acc[5] += h.z;
      }
#pragma unroll
      for (int j = 0; j < KP; ++j)
        if ((bits >> j) & 1) acc[6 + j] += o.Jp[0][j] * g0 + o.Jp[1][j] * g1;
    }
    wave_allsum<6 + KP>(acc);
    if (!cam_seg_total<6 + KP>(g.g, sg, acc, lane)) continue;
    if (lane == 0) {
      double dn = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double wv = acc[j] + yscale * dvec[6 * (long)n + j] * zp[j];
        v.w[6 * (long)n + j] = wv;
        dn += zp[j] * wv;
      }
      delta += dn;
#pragma unroll
      for (int j = 0; j < KP; ++j) yi_part[KP * (long)n + j] = acc[6 + j];
    }
  }
  if (lane == 0) sdelta[wid] = delta;
  __syncthreads();
  if (threadIdx.x == 0) v.dpart[dslot0 + blockIdx.x] = (sdelta[0] + sdelta[1]) + (sdelta[2] + sdelta[3]);
}

// Intrinsics rows of w: one block per intrinsics block sums its cameras' shares (fixed order), adds the
// damping term and this block's part of delta = z.w into dpart[slot0 + blockIdx].
__global__ void __launch_bounds__(kBlock)
    k_ba_phaseI(BaDev g, CgVec v, double yscale, const double* __restrict__ yi_part,
                const double* __restrict__ dvec, int slot0) {
  __shared__ double smem[4 * KP];
  if (v.st->done) return;
  double delta = 0.0;
  for (int k = blockIdx.x; k < g.K; k += gridDim.x) {
    double acc[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) acc[j] = 0.0;
    for (int i = g.ioff[k] + threadIdx.x; i < g.ioff[k + 1]; i += blockDim.x) {
      const double* s = yi_part + KP * (long)g.icams[i];
#pragma unroll
      for (int j = 0; j < KP; ++j) acc[j] += s[j];
    }
    block_sum<KP>(acc, smem);
    if (threadIdx.x == 0) {
      const FreeBits bits = g.intr_free[k];
      const long o = 6 * (long)g.g.N + KP * (long)k;
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        const double zj = v.z[o + j];
        const double wv = (((bits >> j) & 1) ? acc[j] : 0.0) + yscale * dvec[o + j] * zj;
        v.w[o + j] = wv;
        delta += zj * wv;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) v.dpart[slot0 + blockIdx.x] = delta;
}

// One thread per intrinsics block (small groups); block partial of delta via the block tree.
__global__ void __launch_bounds__(kBlock)
    k_ba_phaseI_small(BaDev g, CgVec v, double yscale, const double* __restrict__ yi_part,
                      const double* __restrict__ dvec, int slot0) {
  __shared__ double smem[4];
  if (v.st->done) return;
  double delta[1] = {0.0};
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < g.K; k += gridDim.x * blockDim.x) {
    double acc[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) acc[j] = 0.0;
    for (int i = g.ioff[k]; i < g.ioff[k + 1]; ++i) {
      const double* s = yi_part + KP * (long)g.icams[i];
#pragma unroll
      for (int j = 0; j < KP; ++j) acc[j] += s[j];
    }
    const FreeBits bits = g.intr_free[k];
    const long o = 6 * (long)g.g.N + KP * (long)k;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      const double zj = v.z[o + j];
      const double wv = (((bits >> j) & 1) ? acc[j] : 0.0) + yscale * dvec[o + j] * zj;
      v.w[o + j] = wv;
      delta[0] += zj * wv;
    }
  }
  block_sum<1>(delta, smem);
  if (threadIdx.x == 0) v.dpart[slot0 + blockIdx.x] = delta[0];
}

// ---- the reduced camera system as a DENSE matrix: small problems with chain-like co-visibility ---------------------------
// The counterpart of gp.hip's k_gp_dense_assemble (see there): the reference's mapper on a 300-image ring looking outward spends
// 5 500 joint-block PCG iterations on 11 LM steps of a bundle adjustment with 1 816 reduced unknowns.  With the stored planes
// U_k = [A_k | I_k] (2 x (6 + F), scaled by sqrt(w)) and B_k (2 x 3) of every observation,
//     S = sum over tracks p, observations k, k' of p:   U_k^T (delta_kk' I_2 - B_k H_pp^-1 B_k'^T) U_k'   + D
// in the unknown layout of the PCG ([6 per camera | KP per intrinsics block]).  One workgroup per camera n owns the six pose rows
// (LDS, 6 x ld doubles) and, for the intrinsics rows of the camera's block, the intrinsics COLUMNS only (KP x KP K doubles, added to
// the global matrix by atomics at the end); the intrinsics rows' pose columns are the transpose of what the pose rows' owners
// computed (k_ba_dense_finish, which also adds the damping and puts 1 on the diagonal of unknowns nothing touches: the constant
// camera, parameters that are not optimised, the padding).  Trivial frames, the 8-wide unit, at most kBaDenseMaxUnknowns reduced
// unknowns (assembled in column windows of kBaDenseWindow) and kBaDenseMaxIntr intrinsics blocks, one rank; switched on like GP's
// (knob gp_dense).
constexpr int kBaDenseMaxUnknowns = 6144;  // (the factorisation is n^3: 1 000 images are ~10 ms per solve, still six times the PCG's pace)
constexpr int kBaDenseWindow = 3072;       // columns per assembly window: 6 x 3072 doubles of LDS per workgroup (144 KB) + the intrinsics rows
constexpr int kBaDenseMaxIntr = 16;
constexpr int kBaDenseTrigger = 100;

__global__ void __launch_bounds__(kBlock)
    k_ba_dense_assemble(BaDev g, const double2* __restrict__ jt, const double* __restrict__ pth, int ld, int cw, double* __restrict__ S) {
  // blockIdx.y: the window of cw columns this workgroup accumulates (a row of ld doubles x 6 does not fit the LDS beyond
  // ld = 3072; wider systems are assembled in several column windows, every window walking the observations again)
  extern __shared__ double srow[];  // [6][cw] | [KP][KP K]
  const int n = blockIdx.x, N = g.g.N, F = g.F, nik = KP * g.K;
  const int c0 = blockIdx.y * cw, c1 = min(ld, c0 + cw);
  const bool first = blockIdx.y == 0;  // (the intrinsics rows' intrinsics columns: once)
  double* sint = srow + 6 * (size_t)cw;
  for (int i = threadIdx.x; i < 6 * cw + KP * nik; i += blockDim.x) srow[i] = 0.0;
  __syncthreads();
  const int ikn = g.cam_intr[n];
  const Map8 mpn = load_map(g.intr_map + KP * (long)ikn);
  for (int slot = g.g.coff[n] + threadIdx.x; slot < g.g.coff[n + 1]; slot += blockDim.x) {
    const long k = g.g.c_src[slot];
    const long p = g.g.c_pt[slot];
    double2 ak[6], ik[KP], bk[3];
#pragma unroll
    for (int j = 0; j < 6; ++j) ak[j] = jt[(PL_A + j) * g.Mp + k];
#pragma unroll
    for (int j = 0; j < KP; ++j) ik[j] = j < F ? jt[(PL_I + j) * g.Mp + k] : make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < 3; ++j) bk[j] = jt[(PL_B + j) * g.Mp + k];
    const double* h = pth + 6 * p;
    const S3 Hi{h[0], h[1], h[2], h[3], h[4], h[5]};
    // M = B_k H_pp^-1 (2 x 3), row r = H_pp^-1 (B_k row r)^T
    const V3 m0 = mul(Hi, V3{bk[0].x, bk[1].x, bk[2].x}), m1 = mul(Hi, V3{bk[0].y, bk[1].y, bk[2].y});
    for (long m = g.g.off[p]; m < g.g.off[p + 1]; ++m) {
      const int j = g.g.cam[m];
      const int ikj = g.obs_ik[m];
      const double2 b0 = jt[(PL_B + 0) * g.Mp + m], b1 = jt[(PL_B + 1) * g.Mp + m], b2 = jt[(PL_B + 2) * g.Mp + m];
      const double d = m == k ? 1.0 : 0.0;
      // W = delta I - M B_m^T
      const double w00 = d - (m0.x * b0.x + m0.y * b1.x + m0.z * b2.x), w01 = -(m0.x * b0.y + m0.y * b1.y + m0.z * b2.y);
      const double w10 = -(m1.x * b0.x + m1.y * b1.x + m1.z * b2.x), w11 = d - (m1.x * b0.y + m1.y * b1.y + m1.z * b2.y);
      const Map8 mpj = load_map(g.intr_map + KP * (long)ikj);
      // T = W U_m, column by column; pose rows += A_k^T T, intrinsics rows (their intrinsics columns) += I_k^T T
      for (int c = 0; c < 6 + F; ++c) {
        const double2 u = c < 6 ? jt[(PL_A + c) * g.Mp + m] : jt[(PL_I + (c - 6)) * g.Mp + m];
        int col;
        if (c < 6) {
          col = 6 * j + c;
        } else {
          const int pm = mpj.m[c - 6];
          if (pm < 0) continue;
          col = 6 * N + KP * ikj + pm;
        }
        const double t0 = w00 * u.x + w01 * u.y, t1 = w10 * u.x + w11 * u.y;
        if (col >= c0 && col < c1) {
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            const double v = ak[a].x * t0 + ak[a].y * t1;
            if (v != 0.0) atomicAdd(srow + (size_t)a * cw + (col - c0), v);
          }
        }
        if (c >= 6 && first) {
#pragma unroll
          for (int a = 0; a < KP; ++a) {
            const int pa = mpn.m[a];
            if (a < F && pa >= 0) {
              const double v = ik[a].x * t0 + ik[a].y * t1;
              if (v != 0.0) atomicAdd(sint + (size_t)pa * nik + (col - 6 * N), v);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 6 * cw; i += blockDim.x)
    if (c0 + i % cw < c1) S[((size_t)6 * n + i / cw) * ld + c0 + i % cw] = srow[i];
  if (first)
  for (int i = threadIdx.x; i < KP * nik; i += blockDim.x) {
    const double v = sint[i];
    if (v != 0.0) unsafeAtomicAdd(S + ((size_t)6 * N + KP * ikn + i / nik) * ld + 6 * N + i % nik, v);
  }
}
// rows >= 6 N: the pose columns of the intrinsics rows by symmetry; every row: + D, 1 on an empty diagonal; the padding: identity
__global__ void __launch_bounds__(kBlock)
    k_ba_dense_finish(int N, int n_a, int ld, const double* __restrict__ dvec, double* __restrict__ S) {
  const size_t total = (size_t)ld * ld;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld), c = (int)(i % ld);
    if (r >= n_a || c >= n_a) {
      S[i] = r == c ? 1.0 : 0.0;
      continue;
    }
    if (r >= 6 * N && c < 6 * N) S[i] = S[(size_t)c * ld + r];
    if (r == c) {
      const double v = S[i] + dvec[r];
      S[i] = v != 0.0 ? v : 1.0;
    }
  }
}

// ---- back-substitution, model cost change, candidate points ------------------------------------
// One lane per observation over the stored planes: dX_p = -e_p - H_pp^-1 sum_k B_k^T u_k (segmented wave
// scan), broadcast back to the track's lanes for the model decrease  -sum_k (m.rw + m.m / 2),
// m = u_k + B_k dX.  part[block][3] = {model_cost_change, |dX|^2, |X|^2}
template <int F>
__device__ __forceinline__ void ba_obs_u(const BaDev& g, const double2* __restrict__ jt, const double* __restrict__ dv,
                                         const double* __restrict__ dintr, long k, double& u0, double& u1) {
  const long n = g.g.cam[k];
  const double2* vp = reinterpret_cast<const double2*>(dv + 6 * n);
  const double2 v01 = vp[0], v23 = vp[1], v45 = vp[2];
  const double vv[6] = {v01.x, v01.y, v23.x, v23.y, v45.x, v45.y};
  u0 = 0.0;
  u1 = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double2 a = jt[(PL_A + j) * g.Mp + k];
    u0 += a.x * vv[j];
    u1 += a.y * vv[j];
  }
  if constexpr (F > 0) {
    const int ik = g.obs_ik[k];
    const Map8 mp = load_map(g.intr_map + KP * (long)ik);
#pragma unroll
    for (int j = 0; j < F; ++j) {
      const int pm = mp.m[j];
      const double zv = pm >= 0 ? dintr[KP * (long)ik + pm] : 0.0;
      const double2 a = jt[(PL_I + j) * g.Mp + k];
      u0 += a.x * zv;
      u1 += a.y * zv;
    }
  }
}

template <int F>
__global__ void __launch_bounds__(kBlock)
    k_ba_backsub(BaDev g, const double* __restrict__ X, const double2* __restrict__ jt,
                 const double* __restrict__ ptb, const double* __restrict__ dv, double* __restrict__ Xn,
                 double* __restrict__ part) {
  __shared__ double smem[4 * 3];
  double acc3[3] = {0, 0, 0};
  const double* dintr = dv + 6 * (long)g.g.N;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  for (int tile = wave; tile < g.g.T; tile += nwaves) {
    const long k0 = g.g.tile_k[tile], k1 = g.g.tile_k[tile + 1];
    // a tile of at most 64 observations (all but the tiles of tracks longer than a wave): every lane keeps its
    // observation's u and B planes in registers between the per-track sum and the per-observation model term — ONE
    // trip over the Jacobian planes instead of two (same operations in the same order)
    const bool one_trip = k1 - k0 <= 64;
    double acc[3] = {0, 0, 0};
    int key = -1 - lane;
    double ku0 = 0.0, ku1 = 0.0;
    double2 kb0{0, 0}, kb1{0, 0}, kb2{0, 0};
    for (long k = k0 + lane; k < k1; k += 64) {
      key = g.g.obs_pt[k];
      double u0, u1;
      ba_obs_u<F>(g, jt, dv, dintr, k, u0, u1);  // planes of unused tracks are zero
      const double2 b0 = jt[(PL_B + 0) * g.Mp + k], b1 = jt[(PL_B + 1) * g.Mp + k], b2 = jt[(PL_B + 2) * g.Mp + k];
      acc[0] += b0.x * u0 + b0.y * u1;
      acc[1] += b1.x * u0 + b1.y * u1;
      acc[2] += b2.x * u0 + b2.y * u1;
      ku0 = u0;
      ku1 = u1;
      kb0 = b0;
      kb1 = b1;
      kb2 = b2;
    }
    seg_scan<3>(acc, key, lane);
    const bool tail = seg_is_tail(key, lane) && key >= 0;
    V3 dX{0, 0, 0};
    if (tail) {
      const long p = key;
      const V3 Xp = ld3(X + 3 * p);
      if (g.g.used[p] && g.opt_pts) {
        const double* b = ptb + kPtbBa * p;
        dX = V3{0, 0, 0} - ld3(b + 3) - mul(S3{b[6], b[7], b[8], b[9], b[10], b[11]}, V3{acc[0], acc[1], acc[2]});
      }
      st3(Xn + 3 * p, Xp + dX);
      if (g.g.used[p]) {
        acc3[1] += dot(dX, dX);
        acc3[2] += dot(Xp, Xp);
      }
    }
    // broadcast dX of each segment from its tail lane to all its lanes
    const unsigned long long tmask = __ballot(tail);
    const unsigned long long above = tmask >> lane;
    const int src = above ? lane + __ffsll((long long)above) - 1 : lane;
    dX.x = __shfl(dX.x, src, 64);
    dX.y = __shfl(dX.y, src, 64);
    dX.z = __shfl(dX.z, src, 64);
    if (one_trip) {
      const long k = k0 + lane;
      if (k < k1) {
        const double u0 = ku0 + (kb0.x * dX.x + kb1.x * dX.y + kb2.x * dX.z);
        const double u1 = ku1 + (kb0.y * dX.x + kb1.y * dX.y + kb2.y * dX.z);
        const double2 rw = jt[(PL_I + F) * g.Mp + k];
        acc3[0] -= u0 * rw.x + u1 * rw.y + 0.5 * (u0 * u0 + u1 * u1);
      }
    } else {
      for (long k = k0 + lane; k < k1; k += 64) {
        double u0, u1;
        ba_obs_u<F>(g, jt, dv, dintr, k, u0, u1);
        const double2 b0 = jt[(PL_B + 0) * g.Mp + k], b1 = jt[(PL_B + 1) * g.Mp + k], b2 = jt[(PL_B + 2) * g.Mp + k];
        u0 += b0.x * dX.x + b1.x * dX.y + b2.x * dX.z;
        u1 += b0.y * dX.x + b1.y * dX.y + b2.y * dX.z;
        const double2 rw = jt[(PL_I + F) * g.Mp + k];
        acc3[0] -= u0 * rw.x + u1 * rw.y + 0.5 * (u0 * u0 + u1 * u1);
      }
    }
  }
  block_sum<3>(acc3, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) part[blockIdx.x * 3 + k] = acc3[k];
  }
}

// Candidate poses / intrinsics: q' = [cos|d|, sin|d| d/|d|] * q (EigenQuaternionManifold),
// t' = t + dt, par' = par + dpar.  part[block][3] = {|step|^2, |x|^2, #non-finite}.
__global__ void __launch_bounds__(kBlock)
    k_ba_param_update(int N, int K, const double* __restrict__ q, const double* __restrict__ t,
                      const double* __restrict__ par, const double* __restrict__ dv, double* __restrict__ qn,
                      double* __restrict__ tn, double* __restrict__ parn, double* __restrict__ part) {
  __shared__ double smem[4 * 3];
  double acc[3] = {0, 0, 0};
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const double* d = dv + 6 * (long)n;
    const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double s, c;
    sincos(th, &s, &c);
    const double k = th > 0.0 ? s / th : 1.0;
    const Quat qd{c, k * d[0], k * d[1], k * d[2]};
    const Quat q0{q[4 * n], q[4 * n + 1], q[4 * n + 2], q[4 * n + 3]};
    const Quat q1 = qmul(qd, q0);
    qn[4 * n] = q1.w; qn[4 * n + 1] = q1.x; qn[4 * n + 2] = q1.y; qn[4 * n + 3] = q1.z;
    acc[0] += (q1.w - q0.w) * (q1.w - q0.w) + (q1.x - q0.x) * (q1.x - q0.x) + (q1.y - q0.y) * (q1.y - q0.y) +
              (q1.z - q0.z) * (q1.z - q0.z);
    acc[1] += q0.w * q0.w + q0.x * q0.x + q0.y * q0.y + q0.z * q0.z;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double tj = t[3 * n + j];
      tn[3 * n + j] = tj + d[3 + j];
      acc[0] += d[3 + j] * d[3 + j];
      acc[1] += tj * tj;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[2] += isfinite(d[j]) ? 0.0 : 1.0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < KP * K; i += gridDim.x * blockDim.x) {
    const double d = dv[6 * (long)N + i];
    parn[i] = par[i] + d;
    acc[0] += d * d;
    acc[1] += par[i] * par[i];
    acc[2] += isfinite(d) ? 0.0 : 1.0;
  }
  block_sum<3>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) part[blockIdx.x * 3 + k] = acc[k];
  }
}

// candidate cost, one lane per observation; part[block][1]
template <bool WIDE>
__global__ void __launch_bounds__(kBlock)
    k_ba_cost(BaDev g, const double* __restrict__ camR, const double* __restrict__ t,
              const double* __restrict__ X, const double* __restrict__ par, double* __restrict__ part) {
  __shared__ double smem[4];
  double v[1] = {0.0};
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < g.g.M; k += (long)gridDim.x * blockDim.x) {
    const long p = g.g.obs_pt[k];
    if (!g.g.used[p]) continue;
    const int n = g.g.cam[k];
    const int ik = g.cam_intr[n];
    ObsGeom o;
    obs_geom<WIDE, KP>(camR + 9 * (long)n, t + 3 * (long)n, ld3(X + 3 * p), g.intr_model[ik], par + KP * (long)ik, o);
    if (!o.valid) continue;
    const double r0 = o.px - g.xy[2 * k], r1 = o.py - g.xy[2 * k + 1];
    double rho, w;
    huber(g.huber_a, 1.0, r0 * r0 + r1 * r1, rho, w);
    v[0] += 0.5 * rho;
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
// The modes deflated from the reduced solves (CgDeflation in cg.hpp): the similarity gauge of the scene in the
// pose unknowns of the reduced system (6 per camera: left quaternion tangent — the manifold turns by 2 |d| — then
// translation; the intrinsics part of the modes is zero and W was cleared by the caller):
//   world translation a:  dt_n = -R_n a          world rotation w:  drot_n = -R_n w / 2          scale:  dt_n = t_n
// The constant frame has zero entries.  with_rot = 1: [3 translations | 3 rotations | scale] (7 modes);
// with_rot = 0 (rotations frozen): [3 translations | scale] (4 modes).
__global__ void __launch_bounds__(kBlock) k_ba_defl_modes(int N, long n, const double* __restrict__ R, const double* __restrict__ t,
                                                          int fixed_cam, int with_rot, double* __restrict__ W) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N; c += gridDim.x * blockDim.x) {
    if (c == fixed_cam) continue;
    const double* R9 = R + 9L * c;
    const long o = 6L * c;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        W[(size_t)a * n + o + 3 + i] = -R9[3 * i + a];
        if (with_rot) W[(size_t)(3 + a) * n + o + i] = -0.5 * R9[3 * i + a];
      }
    const int js = with_rot ? 6 : 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) W[(size_t)js * n + o + 3 + i] = t[3L * c + i];
  }
}

// ---- A W of the gauge modes in closed form ------------------------------------------------------------------------------
// The deflated modes (k_ba_defl_modes) are exact symmetries of the reprojection error: moving the cameras by W_j and the
// points by m_j(p) — e_a, e_a x X_p, X_p for world translation, rotation, scale — leaves every residual unchanged to first
// order, J_cam W_j + J_pt m_j = 0 per observation.  Hence, with u the unknowns of the reduced system (poses but the constant
// one, intrinsics), F_p the constant camera's share J_pt^T w J_pt of H_p and D_p the point damping,
//     H_uu W + H_up m = -r_u,   H_pu W + H_p m = F_p m      =>      S W = H_uu W - H_up (H_p + D_p)^-1 H_pu W
//                                                                        = -r_u - H_up (H_p + D_p)^-1 (D_p + F_p) m
// where r_u is non-zero only on the intrinsics of the constant camera and equals -sum_k J_i^T w J_pt m there.  Per
// observation of camera n:   (A W_j)_u += J_u^T w J_pt g_j,   g_j = [n is the constant camera] m_j - H_pp^-1 (D_p + F_p) m_j
// (H_pp^-1 = the damped inverse of the build record) — ONE camera-major sweep over the point build records instead of one
// operator application per mode (7 x 340 us per deflated solve at configs[3]).  Needs optimised points (nothing is
// eliminated otherwise) and intrinsics blocks that belong to one camera each (or none free): group sums are not formed here.

// F_p of the points the constant camera sees: one thread per observation of that camera (camera-major slots
// [k0, k1)); slot = k - k0, recorded in the point's build record.
template <bool WIDE>
__global__ void __launch_bounds__(kBlock)
    k_ba_fixed_share(BaDev g, const int* __restrict__ slots, int ns, const double* __restrict__ camR, const double* __restrict__ t,
                     const double* __restrict__ par, const double* __restrict__ c_w, double* __restrict__ ptb,
                     double* __restrict__ ftab) {
  const int n = g.fixed_cam;
  const int ik = g.cam_intr[n];
  const double* R9 = camR + 9 * (long)n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const int k = slots[i];
    double* b = ptb + kPtbBa * (long)g.g.c_pt[k];
    const double w = c_w[k];
    ObsGeom o;
    obs_geom<WIDE, KP>(R9, t + 3 * (long)n, ld3(b), g.intr_model[ik], par + KP * (long)ik, o);
    ObsJac J;
    build_jac(g, n, R9, o, J);
    double* f = ftab + 6 * (long)i;
    f[0] = w * (J.Jpt[0][0] * J.Jpt[0][0] + J.Jpt[1][0] * J.Jpt[1][0]);
    f[1] = w * (J.Jpt[0][0] * J.Jpt[0][1] + J.Jpt[1][0] * J.Jpt[1][1]);
    f[2] = w * (J.Jpt[0][0] * J.Jpt[0][2] + J.Jpt[1][0] * J.Jpt[1][2]);
    f[3] = w * (J.Jpt[0][1] * J.Jpt[0][1] + J.Jpt[1][1] * J.Jpt[1][1]);
    f[4] = w * (J.Jpt[0][1] * J.Jpt[0][2] + J.Jpt[1][1] * J.Jpt[1][2]);
    f[5] = w * (J.Jpt[0][2] * J.Jpt[0][2] + J.Jpt[1][2] * J.Jpt[1][2]);
    b[15] = (double)i;
  }
}

// (the lean instance with two free intrinsics columns takes 272 registers left alone = ONE wave per SIMD; bounded to 256 it
// spills 18 and runs 375 -> 286 us at configs[3]: profiles/r05_ab_small_experiments.txt.  The same medicine made
// k_gp_build_cam<LIN, AW> (169 -> 162, third wave) and k_ba_lin_track (172 -> 162, third wave) SLOWER: 374 -> 404, 487 -> 510 us.)
template <bool ROT, bool WIDE, int F>
__global__ void __launch_bounds__(kBlock, (!WIDE && F <= 2) ? 2 : 1)
    k_ba_aw_modes(BaDev g, double yscale, const double* __restrict__ camR, const double* __restrict__ t,
                  const double* __restrict__ par, const double* __restrict__ c_w, const double* __restrict__ ptb,
                  const double* __restrict__ ftab, const double* __restrict__ dvec, const double* __restrict__ W,
                  double* __restrict__ AW, long nvec,
                  double* __restrict__ ipart /* [N][NM][KP]: the camera's share of its (shared) intrinsics block's rows, or null */) {
  constexpr int NM = ROT ? 7 : 4;  // [3 translations | 3 rotations | scale] or [3 translations | scale]
  constexpr int U = 6 + F;
  constexpr int NACC = NM * U;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  for (int it = wave; it < cam_seg_count(g.g); it += nwaves) {
    const int sg = cam_seg_index(g.g, it);
    const int n = g.g.seg_cam[sg];
    const int ik = g.cam_intr[n];
    const int model = g.intr_model[ik];
    const double* R9 = camR + 9 * (long)n;
    const double* t3 = t + 3 * (long)n;
    const double* pp = par + KP * (long)ik;
    const Map8 mp = load_map(g.intr_map + KP * (long)ik);
    const double own = n == g.fixed_cam ? 1.0 : 0.0;
    double acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.0;
    for (int k = cam_seg_k0(g.g, sg) + lane; k < cam_seg_k1(g.g, sg); k += 64) {
      const double w = c_w[k];
      if (w == 0.0) continue;
      const double* b = ptb + kPtbBa * (long)g.g.c_pt[k];
      const V3 Xp = ld3(b);
      const S3 Hi{b[6], b[7], b[8], b[9], b[10], b[11]};
      S3 B{b[12], 0.0, 0.0, b[13], 0.0, b[14]};  // D_p + F_p
      const int slot = (int)b[15];
      if (slot >= 0) {
        const double* f = ftab + 6 * (long)slot;
        B.xx += f[0]; B.xy += f[1]; B.xz += f[2]; B.yy += f[3]; B.yz += f[4]; B.zz += f[5];
      }
      ObsGeom o;
      obs_geom<WIDE, KP>(R9, t3, Xp, model, pp, o);
      ObsJac J;
      build_jac(g, n, R9, o, J);
      double Jc[2][F > 0 ? F : 1];
#pragma unroll
      for (int j = 0; j < F; ++j) {
        const int pm = mp.m[j];
        Jc[0][j] = sel8(o.Jp[0], pm);
        Jc[1][j] = sel8(o.Jp[1], pm);
      }
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        V3 m;
        if (j < 3) {
          m = V3{j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
        } else if (ROT && j < 6) {  // e_a x X_p
          const int a = j - 3;
          m = a == 0 ? V3{0.0, -Xp.z, Xp.y} : (a == 1 ? V3{Xp.z, 0.0, -Xp.x} : V3{-Xp.y, Xp.x, 0.0});
        } else {
          m = Xp;
        }
        const V3 gj = own * m - mul(Hi, mul(B, m));
        const double q0 = w * (J.Jpt[0][0] * gj.x + J.Jpt[0][1] * gj.y + J.Jpt[0][2] * gj.z);
        const double q1 = w * (J.Jpt[1][0] * gj.x + J.Jpt[1][1] * gj.y + J.Jpt[1][2] * gj.z);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[j * U + i] += J.Jpose[0][i] * q0 + J.Jpose[1][i] * q1;
#pragma unroll
        for (int i = 0; i < F; ++i) acc[j * U + 6 + i] += Jc[0][i] * q0 + Jc[1][i] * q1;
      }
    }
    wave_allsum<NACC>(acc);
    if (!cam_seg_total<NACC>(g.g, sg, acc, lane)) continue;
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        double* out = AW + (size_t)j * nvec;
        const double* wj = W + (size_t)j * nvec;
#pragma unroll
        for (int i = 0; i < 6; ++i) out[6 * (long)n + i] = acc[j * U + i] + yscale * dvec[6 * (long)n + i] * wj[6 * (long)n + i];
        if (ipart == nullptr) {  // the block is this camera's own
#pragma unroll
          for (int i = 0; i < F; ++i) {
            const int pm = mp.m[i];
            if (pm >= 0) out[6 * (long)g.g.N + KP * (long)ik + pm] = acc[j * U + 6 + i];
          }
        } else {  // a block shared by several cameras: the shares are added per block in camera order (k_ba_group_sum)
          double* ip = ipart + ((size_t)n * NM + j) * KP;
#pragma unroll
          for (int i = 0; i < KP; ++i) ip[i] = 0.0;
#pragma unroll
          for (int i = 0; i < F; ++i) {
            const int pm = mp.m[i];
            if (pm >= 0) ip[pm] = acc[j * U + 6 + i];
          }
        }
      }
    }
  }
}

// AW[j][6 N + 8 k + i] = gsum[k][j][i]: the per-block sums of the cameras' shares (shared intrinsics blocks)
__global__ void __launch_bounds__(kBlock)
    k_ba_aw_intr_scatter(int N, int K, int nm, const double* __restrict__ gsum, double* __restrict__ AW, long nvec) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)K * nm * KP; i += (long)gridDim.x * blockDim.x) {
    const long k = i / (nm * KP);
    const int j = (int)((i / KP) % nm), c = (int)(i % KP);
    AW[(size_t)j * nvec + 6 * (long)N + KP * k + c] = gsum[i];
  }
}

// flag[0] = 1 when two consecutive camera-major slots in [k0, k1) name the same point (a camera's list is in track order)
__global__ void __launch_bounds__(kBlock) k_ba_dup_check(const int* __restrict__ c_pt, const int* __restrict__ slots, int ns,
                                                         int* __restrict__ flag) {
  for (int i = 1 + blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x)
    if (c_pt[slots[i]] == c_pt[slots[i - 1]]) atomicOr(flag, 1);  // (the list is in track order)
}

struct BaWs {
  ObsGraphWs og;
  DevBuf<long> off;
  DevBuf<int> cam, cam_intr, intr_model, ioff, icams, obs_ik, fix_slots;
  DevBuf<FreeBits> intr_free;
  DevBuf<signed char> intr_map, intr_slot;
  DevBuf<double2> jt;
  DevBuf<double> xy, c_xy, c_w, q, qn, t, tn, camR, camRn, X, Xn, par, parn, ptH, ptb, pth, ptrec, ptdiag, ptjs, diag, js,
      awi_part, awi_sum, dvec, grad, gred, rhs, spose, scross, minvj, zrec, ipart, iacc16, iacc44, yi_part, minv, cg_x, cg_r, cg_z, cg_p, cg_s, cg_w, vpart,
      dpart, part, scal;
  DevBuf<CgStatus> cgst;
  DevBuf<CgScal> cgsc;
  // calibrated rigs: image tables and the image-space twins of the per-camera arrays
  DevBuf<int> img_frame, foff, fimg, img_sensor, soff, simg;
  DevBuf<unsigned char> img_fixed, fmask;
  DevBuf<double> sens, Ri, Rin, ti, tin, diag_i, grad_i, gred_i, spose_i, dvec_i, zimg, wimg, ximg, lever, gram_i;
  DevBuf<double> defl_w, defl_aw, defl_b2, defl_part, defl_small, defl_cd;  // CgDeflation, cg.hpp
  DevBuf<double> ftab;  // [observations of the constant camera][6]: its share of H_pp (k_ba_fixed_share)
  DevBuf<double> dn_S, dn_a, dn_b, dn_pinv, dn_r, dn_dx, dn_sc, dn_nrm;  // dense reduced system (k_ba_dense_*)
  DevBuf<double> maxpart;
  static void destroy(void* p) { delete static_cast<BaWs*>(p); }
};

BaWs* ba_ws(gsfm_ctx* ctx) {  // one workspace per unit: BaWs is this unit's own type
  void*& slot = KP == 8 ? ctx->ba_ws : ctx->ba_ws_wide;
  if (!slot) {
    slot = new BaWs();
    (KP == 8 ? ctx->ba_ws_free : ctx->ba_ws_wide_free) = &BaWs::destroy;
  }
  return static_cast<BaWs*>(slot);
}

int num_params_of(int model) {
  switch (model) {
    case GSFM_CAMERA_SIMPLE_PINHOLE: return 3;
    case GSFM_CAMERA_PINHOLE: return 4;
    case GSFM_CAMERA_SIMPLE_RADIAL: return 4;
    case GSFM_CAMERA_RADIAL: return 5;
    case GSFM_CAMERA_OPENCV: return 8;
    case GSFM_CAMERA_OPENCV_FISHEYE: return 8;
    case GSFM_CAMERA_FOV: return 5;
    case GSFM_CAMERA_SIMPLE_RADIAL_FISHEYE: return 4;
    case GSFM_CAMERA_RADIAL_FISHEYE: return 5;
    // more than 8 parameters: the 16-wide unit only (ba_wide.hip; gsfm_ba_problem::intr_stride = 16)
    case GSFM_CAMERA_FULL_OPENCV: return KP >= 16 ? 12 : -1;
    case GSFM_CAMERA_THIN_PRISM_FISHEYE: return KP >= 16 ? 12 : -1;
    case GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE: return KP >= 16 ? 16 : -1;
    default: return -1;
  }
}
unsigned pp_mask_of(int model) {
  switch (model) {
    case GSFM_CAMERA_PINHOLE:
    case GSFM_CAMERA_OPENCV:
    case GSFM_CAMERA_OPENCV_FISHEYE:
    case GSFM_CAMERA_FOV:
    case GSFM_CAMERA_FULL_OPENCV:
    case GSFM_CAMERA_THIN_PRISM_FISHEYE:
    case GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE: return (1u << 2) | (1u << 3);
    default: return (1u << 1) | (1u << 2);
  }
}

// Launch of a kernel that projects: the instance without the fisheye / FOV branches unless a camera needs them
// (camera.hpp: distort_project<WIDE>).  `wide_` is the solver's flag.
// (The 16-wide unit has the WIDE instance only: its three models live in distort_project_wide.)
#define WIDE_LAUNCH(kernel, ...)                                    \
  do {                                                              \
    if (KP > 8 || wide_) {                                          \
      constexpr bool WIDE = true;                                   \
      hipLaunchKernelGGL(kernel, __VA_ARGS__);                      \
    } else {                                                        \
      constexpr bool WIDE = KP > 8;                                 \
      hipLaunchKernelGGL(kernel, __VA_ARGS__);                      \
    }                                                               \
  } while (0)

// F = stored free-intrinsics columns: 0, 2, 4 or the full width (KP = 8); 0 or the full width (KP = 16)
template <typename Fn>
void dispatch_f(int F, Fn&& fn) {
  if constexpr (KP == 8) {
    switch (F) {
      case 0: fn(std::integral_constant<int, 0>{}); break;
      case 2: fn(std::integral_constant<int, 2>{}); break;
      case 4: fn(std::integral_constant<int, 4>{}); break;
      default: fn(std::integral_constant<int, 8>{}); break;
    }
  } else {
    if (F == 0) fn(std::integral_constant<int, 0>{}); else fn(std::integral_constant<int, KP>{});
  }
}

// ---- calibrated rigs -------------------------------------------------------------------------------------------
// colmap::RigReprojErrorConstantRigCostFunctor (bundle_adjustment.cc:147-160): x_c = R_s (R_f X + t_f) + t_s with a
// constant cam_from_rig (R_s, t_s) per image.  Every sweep above keeps working on "cameras" = IMAGES with the composed
// pose (R_s R_f, R_s t_f + t_s) and its own left-multiplicative tangent; the unknown is the FRAME's pose, whose tangent
// (d_rot, d_trn) maps to the image's as T_s d = (R_s d_rot, R_s d_trn)  [R_s [a]x R_s^T = [R_s a]x].  So the LM diagonal,
// the block-Jacobi blocks and the PCG vectors live per frame, and small kernels translate around the sweeps:
// z_image = T_s z_frame before them, w_frame = sum_images T_s^T w_image after them.  z_frame . w_frame =
// sum z_image . w_image (adjoint maps): the delta partials of the image-space sweeps are the frame-space ones.
//
// colmap::RigReprojErrorCostFunctor (bundle_adjustment.cc:161-179, optimize_rig_poses): the cam_from_rig of every
// non-reference sensor is a parameter block of its own, stored as pose block N + s behind the N frames (same manifold,
// same update kernel).  Its tangent (d_rot, d_trn) moves the image as
//   R_i <- Exp(2 d_rot) R_s R_f,   t_i <- Exp(2 d_rot) (R_s t_f) + t_s + d_trn
// i.e. U_i d = (d_rot, d_trn - 2 b x d_rot) with the lever b = R_s t_f, so z_image = T_s z_frame + U_i z_sensor and
// w_sensor = sum_images U_i^T w_image, U^T w = (w_rot + 2 b x w_trn, w_trn).  The image Jacobians stay unmasked in this
// mode (a sensor block is never constant, ba.cc:296-309) and the constant frame / optimize_rotations /
// optimize_translation act on the frame rows instead (fmask).
struct RigDev {
  int NI, N, S;                // images, frames, optimised sensor blocks (pose blocks N .. N + S - 1)
  const int* img_frame;        // [NI]
  const int* img_sensor;       // [NI] sensor block of the image or -1; null when S == 0
  double* sens;                // [NI][12] cam_from_rig (R row-major | t) at the linearisation point
  double* lever;               // [NI][3] R_s t_frame (S > 0)
  const int* foff;             // [N + 1] frame -> images (ascending: a fixed summation order)
  const int* fimg;             // [NI]
  const int* soff;             // [S + 1] sensor block -> images
  const int* simg;
  const unsigned char* fmask;  // [N] bit 0: rotation free, bit 1: translation free; null: the masks live in the image Jacobians
};

__device__ __forceinline__ void rig_sensor_of(const RigDev& rg, int i, const double* __restrict__ Rp,
                                              const double* __restrict__ tp, double (&S)[12]) {
  const int sb = rg.img_sensor ? rg.img_sensor[i] : -1;
  if (sb >= 0) {
#pragma unroll
    for (int j = 0; j < 9; ++j) S[j] = Rp[9 * (long)(rg.N + sb) + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) S[9 + j] = tp[3 * (long)(rg.N + sb) + j];
  } else {
#pragma unroll
    for (int j = 0; j < 12; ++j) S[j] = rg.sens[12 * (long)i + j];
  }
}

// image poses = cam_from_rig * rig_from_world for the pose blocks (Rp, tp): frames, then the optimised sensors
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_poses(RigDev rg, const double* __restrict__ Rp, const double* __restrict__ tp, double* __restrict__ Ri,
                   double* __restrict__ ti) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rg.NI; i += gridDim.x * blockDim.x) {
    double S[12];
    rig_sensor_of(rg, i, Rp, tp, S);
    const double* R = Rp + 9 * (long)rg.img_frame[i];
    const double* t = tp + 3 * (long)rg.img_frame[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) Ri[9 * (long)i + 3 * a + b] = S[3 * a] * R[b] + S[3 * a + 1] * R[3 + b] + S[3 * a + 2] * R[6 + b];
      ti[3 * (long)i + a] = S[3 * a] * t[0] + S[3 * a + 1] * t[1] + S[3 * a + 2] * t[2] + S[9 + a];
    }
  }
}

// sensor mode, at every linearisation point: the images' cam_from_rig table and levers from the current pose blocks
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_refresh(RigDev rg, const double* __restrict__ Rp, const double* __restrict__ tp) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rg.NI; i += gridDim.x * blockDim.x) {
    double S[12];
    rig_sensor_of(rg, i, Rp, tp, S);
    const double* t = tp + 3 * (long)rg.img_frame[i];
#pragma unroll
    for (int j = 0; j < 12; ++j) rg.sens[12 * (long)i + j] = S[j];
#pragma unroll
    for (int a = 0; a < 3; ++a) rg.lever[3 * (long)i + a] = S[3 * a] * t[0] + S[3 * a + 1] * t[1] + S[3 * a + 2] * t[2];
  }
}

// dst_image = T_s src_frame (+ U_i src_sensor) | intrinsics part copied: z and the step dy.  src is laid out
// [6 per pose block (Np of them) | 8 per intrinsics block], dst [6 per image | 8 per intrinsics block].
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_expand(RigDev rg, int Np, int K, const double* __restrict__ src, double* __restrict__ dst) {
  const int total = rg.NI + K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (i < rg.NI) {
      const double* S = rg.sens + 12 * (long)i;
      const double* v = src + 6 * (long)rg.img_frame[i];
      double o[6];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int a = 0; a < 3; ++a) o[3 * h + a] = S[3 * a] * v[3 * h] + S[3 * a + 1] * v[3 * h + 1] + S[3 * a + 2] * v[3 * h + 2];
      const int sb = rg.img_sensor ? rg.img_sensor[i] : -1;
      if (sb >= 0) {
        const double* d = src + 6 * (long)(rg.N + sb);
        const double* b = rg.lever + 3 * (long)i;
        o[0] += d[0];
        o[1] += d[1];
        o[2] += d[2];
        o[3] += d[3] - 2.0 * (b[1] * d[2] - b[2] * d[1]);
        o[4] += d[4] - 2.0 * (b[2] * d[0] - b[0] * d[2]);
        o[5] += d[5] - 2.0 * (b[0] * d[1] - b[1] * d[0]);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) dst[6 * (long)i + j] = o[j];
    } else {
      const int k = i - rg.NI;
#pragma unroll
      for (int j = 0; j < KP; ++j) dst[6 * (long)rg.NI + KP * (long)k + j] = src[6 * (long)Np + KP * (long)k + j];
    }
  }
}

// plain fixed-order sums over the images of a frame (the quantities lin_cam already produced in the frame tangent)
template <int W>
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_sum(int N, const int* __restrict__ foff, const int* __restrict__ fimg, const double* __restrict__ src,
                 double* __restrict__ dst) {
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < N; f += gridDim.x * blockDim.x) {
    double acc[W];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] = 0.0;
    for (int a = foff[f]; a < foff[f + 1]; ++a) {
      const double* sp = src + (long)W * fimg[a];
#pragma unroll
      for (int j = 0; j < W; ++j) acc[j] += sp[j];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) dst[(long)W * f + j] = acc[j];
  }
}

__device__ __forceinline__ void unpack_sym6(const double* __restrict__ sp, double (&A)[6][6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) A[i][j] = A[j][i] = sp[sym6(i, j)];
}

// B += V^T A V for a 6 x 6 tangent map V (row-major)
__device__ __forceinline__ void congruence6(const double (&A)[6][6], const double (&V)[6][6], double (&B)[6][6]) {
  double M[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double m = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) m += A[i][k] * V[k][j];
      M[i][j] = m;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double m = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) m += V[k][i] * M[k][j];
      B[i][j] += m;
    }
}

// U_i = [[I, 0], [-2 [b]x, I]]
__device__ __forceinline__ void sensor_map(const double* __restrict__ b, double (&U)[6][6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) U[i][j] = i == j ? 1.0 : 0.0;
  U[3][1] = 2.0 * b[2];
  U[3][2] = -2.0 * b[1];
  U[4][0] = -2.0 * b[2];
  U[4][2] = 2.0 * b[0];
  U[5][0] = 2.0 * b[1];
  U[5][1] = -2.0 * b[0];
}

// T_s = diag(R_s, R_s)
__device__ __forceinline__ void frame_map(const double* __restrict__ S, double (&T)[6][6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) T[i][j] = (i / 3 == j / 3) ? S[3 * (i % 3) + (j % 3)] : 0.0;
}

// g_out += V^T g
__device__ __forceinline__ void map_t6(const double (&V)[6][6], const double* __restrict__ gi, double (&g)[6]) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double m = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) m += V[k][j] * gi[k];
    g[j] += m;
  }
}

__device__ __forceinline__ bool frame_free(const RigDev& rg, int f, int j) {
  return rg.fmask == nullptr || ((rg.fmask[f] >> (j / 3)) & 1);
}

// sensor mode: squared column norms and gradient of a frame = masked diagonal of sum T^T G T and sum T^T g over its
// images (G, g: the image's Gram matrix and gradient from k_ba_lin_cam<true>)
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_frame_lin(RigDev rg, const double* __restrict__ gram_i, const double* __restrict__ grad_i,
                       double* __restrict__ diag, double* __restrict__ grad) {
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < rg.N; f += gridDim.x * blockDim.x) {
    double g[6] = {0, 0, 0, 0, 0, 0};
    double B[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) B[i][j] = 0.0;
    for (int a = rg.foff[f]; a < rg.foff[f + 1]; ++a) {
      const int im = rg.fimg[a];
      double A[6][6], T[6][6];
      unpack_sym6(gram_i + 21 * (long)im, A);
      frame_map(rg.sens + 12 * (long)im, T);
      congruence6(A, T, B);
      map_t6(T, grad_i + 6 * (long)im, g);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bool fr = frame_free(rg, f, j);
      diag[6 * (long)f + j] = fr ? B[j][j] : 0.0;
      grad[6 * (long)f + j] = fr ? g[j] : 0.0;
    }
  }
}

// sensor mode: the same for sensor block s (pose block N + s), one workgroup per sensor, fixed-order sums
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_sensor_lin(RigDev rg, const double* __restrict__ gram_i, const double* __restrict__ grad_i,
                        double* __restrict__ diag, double* __restrict__ grad) {
  __shared__ double smem[4 * 12];
  for (int sb = blockIdx.x; sb < rg.S; sb += gridDim.x) {
    double acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = 0.0;
    for (int a = rg.soff[sb] + threadIdx.x; a < rg.soff[sb + 1]; a += blockDim.x) {
      const int im = rg.simg[a];
      double A[6][6], U[6][6], B[6][6], g[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) B[i][j] = 0.0;
      unpack_sym6(gram_i + 21 * (long)im, A);
      sensor_map(rg.lever + 3 * (long)im, U);
      congruence6(A, U, B);
      map_t6(U, grad_i + 6 * (long)im, g);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        acc[j] += B[j][j];
        acc[6 + j] += g[j];
      }
    }
    block_sum<12>(acc, smem);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        diag[6 * (long)(rg.N + sb) + j] = acc[j];
        grad[6 * (long)(rg.N + sb) + j] = acc[6 + j];
      }
    }
    __syncthreads();
  }
}

// reduced gradient and diagonal Schur block of a frame: sum over its images of T_s^T g and T_s^T S T_s
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_reduce_blocks(RigDev rg, const double* __restrict__ gred_i, const double* __restrict__ spose_i,
                           double* __restrict__ gred_f, double* __restrict__ spose_f) {
  const double* sens = rg.sens;
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < rg.N; f += gridDim.x * blockDim.x) {
    double g[6] = {0, 0, 0, 0, 0, 0};
    double B[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) B[i][j] = 0.0;
    for (int a = rg.foff[f]; a < rg.foff[f + 1]; ++a) {
      const int im = rg.fimg[a];
      const double* S = sens + 12 * (long)im;
      const double* gi = gred_i + 6 * (long)im;
      const double* sp = spose_i + 21 * (long)im;
      // T^T g: (R_s^T g_rot, R_s^T g_trn)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[3 * h + j] += S[j] * gi[3 * h] + S[3 + j] * gi[3 * h + 1] + S[6 + j] * gi[3 * h + 2];
      // full 6 x 6 of the image, then T^T A T block by block (3 x 3 blocks: R_s^T A_hk R_s)
      double A[6][6];
      unpack_sym6(sp, A);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          double M1[3][3];  // A_hk R_s
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              M1[i][j] = A[3 * h + i][3 * k] * S[j] + A[3 * h + i][3 * k + 1] * S[3 + j] + A[3 * h + i][3 * k + 2] * S[6 + j];
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) B[3 * h + i][3 * k + j] += S[i] * M1[0][j] + S[3 + i] * M1[1][j] + S[6 + i] * M1[2][j];
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) gred_f[6 * (long)f + j] = frame_free(rg, f, j) ? g[j] : 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j)
        spose_f[21 * (long)f + sym6(i, j)] = (frame_free(rg, f, i) && frame_free(rg, f, j)) ? 0.5 * (B[i][j] + B[j][i]) : 0.0;
  }
}

// sensor mode: the same for the sensor blocks, sum over the sensor's images of U^T g and U^T S U (one workgroup each)
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_sensor_blocks(RigDev rg, const double* __restrict__ gred_i, const double* __restrict__ spose_i,
                           double* __restrict__ gred, double* __restrict__ spose) {
  __shared__ double smem[4 * 27];
  for (int sb = blockIdx.x; sb < rg.S; sb += gridDim.x) {
    double g[6] = {0, 0, 0, 0, 0, 0};
    double B[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) B[i][j] = 0.0;
    for (int a = rg.soff[sb] + threadIdx.x; a < rg.soff[sb + 1]; a += blockDim.x) {
      const int im = rg.simg[a];
      double A[6][6], U[6][6];
      unpack_sym6(spose_i + 21 * (long)im, A);
      sensor_map(rg.lever + 3 * (long)im, U);
      congruence6(A, U, B);
      map_t6(U, gred_i + 6 * (long)im, g);
    }
    double acc[27];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = g[j];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) acc[6 + i * 6 - (i * (i - 1)) / 2 + (j - i)] = 0.5 * (B[i][j] + B[j][i]);
    block_sum<27>(acc, smem);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j) gred[6 * (long)(rg.N + sb) + j] = acc[j];
#pragma unroll
      for (int j = 0; j < 21; ++j) spose[21 * (long)(rg.N + sb) + j] = acc[6 + j];
    }
    __syncthreads();
  }
}

// image-space diagonal for the sweeps: zero for the pose columns (their damping is a frame-space term), the
// frame-space values for the intrinsics columns
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_dvec(int NI, int Np, int K, const double* __restrict__ dvec_f, double* __restrict__ dvec_i) {
  const long total = 6 * (long)NI + KP * (long)K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    dvec_i[i] = i < 6 * (long)NI ? 0.0 : dvec_f[6 * (long)Np + (i - 6 * (long)NI)];
}

// w_frame (pose) = sum_images T_s^T w_image + D z_frame, w_sensor = sum_images U_i^T w_image + D z_sensor; intrinsics rows
// copied; the damping share of delta goes to one partial slot per block.  Blocks [0, gf): frames and the intrinsics copy;
// block gf + sb: sensor block sb.
// v is the frame-space vector set: v.N = pose blocks (frames + sensor blocks).
__global__ void __launch_bounds__(kBlock)
    k_ba_rig_reduce_w(CgVec v, RigDev rg, double yscale, const double* __restrict__ w_img, const double* __restrict__ dvec,
                      int dslot, int gf) {
  __shared__ double smem[4 + 4 * 6];
  if (v.st->done) return;
  const double* sens = rg.sens;
  double d[1] = {0.0};
  if ((int)blockIdx.x >= gf) {  // one block per sensor block
    const int sb = blockIdx.x - gf;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int a = rg.soff[sb] + threadIdx.x; a < rg.soff[sb + 1]; a += blockDim.x) {
      const int im = rg.simg[a];
      const double* wi = w_img + 6 * (long)im;
      const double* b = rg.lever + 3 * (long)im;
      acc[0] += wi[0] + 2.0 * (b[1] * wi[5] - b[2] * wi[4]);
      acc[1] += wi[1] + 2.0 * (b[2] * wi[3] - b[0] * wi[5]);
      acc[2] += wi[2] + 2.0 * (b[0] * wi[4] - b[1] * wi[3]);
      acc[3] += wi[3];
      acc[4] += wi[4];
      acc[5] += wi[5];
    }
    block_sum<6>(acc, smem + 4);
    if (threadIdx.x == 0) {
      const long o = 6 * (long)(rg.N + sb);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double z = v.z[o + j];
        const double dz = yscale * dvec[o + j] * z;
        v.w[o + j] = acc[j] + dz;
        d[0] += z * dz;
      }
      v.dpart[dslot + blockIdx.x] = d[0];
    }
    return;
  }
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gf * blockDim.x;
  for (int f = tid; f < rg.N; f += nth) {  // frames: one thread each
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int a = rg.foff[f]; a < rg.foff[f + 1]; ++a) {
      const int im = rg.fimg[a];
      const double* S = sens + 12 * (long)im;
      const double* wi = w_img + 6 * (long)im;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[3 * h + j] += S[j] * wi[3 * h] + S[3 + j] * wi[3 * h + 1] + S[6 + j] * wi[3 * h + 2];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double z = v.z[6 * (long)f + j];
      const double dz = yscale * dvec[6 * (long)f + j] * z;
      v.w[6 * (long)f + j] = (frame_free(rg, f, j) ? acc[j] : 0.0) + dz;
      d[0] += z * dz;
    }
  }
  for (int i = tid; i < KP * v.K; i += nth) v.w[6 * (long)v.N + i] = w_img[6 * (long)rg.NI + i];
  block_sum<1>(d, smem);
  if (threadIdx.x == 0) v.dpart[dslot + blockIdx.x] = d[0];
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
    GSFM_REQUIRE(N_ > 0 && K_ > 0 && P_ >= 0 && M_ >= 0, "BA: bad sizes");
    GSFM_REQUIRE((prob->intr_stride == 0 ? GSFM_CAMERA_MAX_PARAMS : prob->intr_stride) == KP, "BA: intr_stride does not match the unit");
    GSFM_REQUIRE(prob->fixed_cam >= -1 && prob->fixed_cam < N_, "BA: fixed_cam out of range");
    // calibrated rigs: the observation graph is over IMAGES (NI_ cameras), the pose unknowns are the N_ frames
    rig_ = prob->num_images > 0;
    NI_ = rig_ ? prob->num_images : N_;
    ni_ = 6 * NI_ + KP * K_;
    std::vector<int> h_imf, h_ims;
    const int num_sensors = rig_ ? prob->num_sensors : 0;
    if (rig_) {
      GSFM_REQUIRE(prob->image_frame && prob->image_cam_from_rig && prob->image_intr, "BA: image tables missing");
      to_host(ctx_, h_imf, prob->image_frame, (size_t)NI_, mem);
      for (int i = 0; i < NI_; ++i) GSFM_REQUIRE(h_imf[i] >= 0 && h_imf[i] < N_, "BA: image_frame out of range");
      GSFM_REQUIRE(num_sensors >= 0, "BA: num_sensors negative");
      if (num_sensors > 0) {
        GSFM_REQUIRE(prob->image_sensor && prob->sensor_cam_from_rig, "BA: sensor tables missing");
        to_host(ctx_, h_ims, prob->image_sensor, (size_t)NI_, mem);
        for (int i = 0; i < NI_; ++i) GSFM_REQUIRE(h_ims[i] >= -1 && h_ims[i] < num_sensors, "BA: image_sensor out of range");
      }
    }
    // optimize_rig_poses (ba.cc:161-179): the sensor blocks are pose blocks N_ .. N_ + S_ - 1 behind the frames
    S_ = opt_.optimize_rig_poses ? num_sensors : 0;
    sens_ = S_ > 0;
    Np_ = N_ + S_;
    n_ = 6 * Np_ + KP * K_;
    std::vector<long> h_off;
    to_host(ctx_, h_off, reinterpret_cast<const long*>(prob->pt_offset), (size_t)P_ + 1, mem);
    GSFM_REQUIRE(h_off[0] == 0 && h_off[P_] == M_, "BA: pt_offset must start at 0 and end at num_obs");
    std::vector<int> h_model, h_ci;
    to_host(ctx_, h_model, prob->intr_model, (size_t)K_, mem);
    to_host(ctx_, h_ci, rig_ ? prob->image_intr : prob->cam_intr, (size_t)NI_, mem);  // intrinsics block per graph camera
    std::vector<FreeBits> h_free(K_);
    std::vector<signed char> h_map(KP * (size_t)K_, -1), h_slot(KP * (size_t)K_, -1);
    int fmax = 0;
    for (int k = 0; k < K_; ++k) {
      const int np = num_params_of(h_model[k]);
      wide_ = wide_ || h_model[k] >= GSFM_CAMERA_OPENCV_FISHEYE;
      if (np < 0)
        throw StatusError(GSFM_ERR_UNSUPPORTED, h_model[k] == GSFM_CAMERA_FULL_OPENCV || h_model[k] == GSFM_CAMERA_THIN_PRISM_FISHEYE ||
                                                        h_model[k] == GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE
                                                    ? "BA: camera models with more than 8 parameters need intr_stride = 16"
                                                    : "BA: camera model not supported");
      unsigned bits = 0;
      // ba.cc:273-293: SubsetManifold on the principal point / constant block / everything free
      if (opt_.optimize_intrinsics || opt_.optimize_principal_point) {
        bits = (1u << np) - 1u;
        if (opt_.optimize_intrinsics && !opt_.optimize_principal_point) bits &= ~pp_mask_of(h_model[k]);
      }
      h_free[k] = (FreeBits)bits;
      int j = 0;
      for (int p = 0; p < KP; ++p)
        if ((bits >> p) & 1) {
          h_slot[KP * (size_t)k + p] = (signed char)j;
          h_map[KP * (size_t)k + j++] = (signed char)p;
        }
      fmax = std::max(fmax, j);
    }
    F_ = fmax == 0 ? 0 : (KP > 8 ? KP : (fmax <= 2 ? 2 : (fmax <= 4 ? 4 : 8)));
    // cameras grouped by intrinsics block (counting sort, stable)
    std::vector<int> h_ioff(K_ + 1, 0), h_icams(NI_);
    for (int n = 0; n < NI_; ++n) {
      GSFM_REQUIRE(h_ci[n] >= 0 && h_ci[n] < K_, "BA: cam_intr out of range");
      h_ioff[h_ci[n] + 1]++;
    }
    max_group_ = 0;
    for (int k = 0; k < K_; ++k) {
      max_group_ = std::max(max_group_, h_ioff[k + 1]);
      h_ioff[k + 1] += h_ioff[k];
    }
    small_groups_ = max_group_ <= 64;
    // one intrinsics block per camera (COLMAP's default for unordered photo collections): pose and
    // intrinsics of a camera are strongly coupled, so they share ONE 14 x 14 block-Jacobi block
    joint_ = KP == 8 && !rig_ && K_ == N_ && max_group_ == 1 && F_ > 0 && !ctx_->knob[GSFM_KNOB_BA_SEPARATE_BLOCKS];
    {
      std::vector<int> fill(h_ioff.begin(), h_ioff.end() - 1);
      for (int n = 0; n < NI_; ++n) h_icams[fill[h_ci[n]]++] = n;
    }
    copy_in(ctx_, ws->off.ensure(P_ + 1), reinterpret_cast<const long*>(prob->pt_offset), (size_t)P_ + 1, mem);
    copy_in(ctx_, ws->cam.ensure(M_ + 1), prob->obs_cam, (size_t)M_, mem);
    copy_in(ctx_, ws->xy.ensure(2 * (size_t)M_ + 2), prob->obs_xy, 2 * (size_t)M_, mem);
    copy_in(ctx_, ws->cam_intr.ensure(NI_), rig_ ? prob->image_intr : prob->cam_intr, (size_t)NI_, mem);
    copy_in(ctx_, ws->intr_model.ensure(K_), prob->intr_model, (size_t)K_, mem);
    copy_in(ctx_, ws->q.ensure(4 * (size_t)Np_), cam_q, 4 * (size_t)N_, mem);
    copy_in(ctx_, ws->t.ensure(3 * (size_t)Np_), cam_t, 3 * (size_t)N_, mem);
    copy_in(ctx_, ws->X.ensure(3 * (size_t)P_ + 3), pt_xyz, 3 * (size_t)P_, mem);
    copy_in(ctx_, ws->par.ensure(KP * (size_t)K_), intr_params, KP * (size_t)K_, mem);
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->intr_free.ensure(K_), h_free.data(), (size_t)K_ * sizeof(FreeBits), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->intr_map.ensure(KP * (size_t)K_ + KP), h_map.data(), KP * (size_t)K_, hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->intr_slot.ensure(KP * (size_t)K_ + KP), h_slot.data(), KP * (size_t)K_, hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->ioff.ensure(K_ + 1), h_ioff.data(), (size_t)(K_ + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->icams.ensure(NI_), h_icams.data(), (size_t)NI_ * sizeof(int), hipMemcpyHostToDevice, s));
    m_used_ = build_obs_graph(ctx_, ws->og, NI_, P_, M_, h_off, ws->off.get(), ws->cam.get(),
                              opt_.min_num_view_per_track /* ba.cc:122 */, g_.g, nullptr);  // syncs the stream
    const long Mu = g_.g.Mu;
    ws->c_xy.ensure(2 * (size_t)M_ + 2);
    hipLaunchKernelGGL((k_og_gather_f64<2>), dim3(grid_for(Mu, kBlock)), dim3(kBlock), 0, s, Mu, g_.g.c_src,
                       ws->xy.get(), ws->c_xy.get());
    ws->c_w.ensure(M_ + 1);
    Mp_ = ((M_ + 63) / 64) * 64;
    ws->jt.ensure((size_t)(10 + F_) * Mp_ + 64);
    // planes of unused tracks stay zero: they contribute nothing and phase A needs no `used` test
    ws->ptrec.ensure(8 * (size_t)P_ + 8);
    GSFM_HIP_CHECK(hipMemsetAsync(ws->ptrec.get(), 0, (8 * (size_t)P_ + 8) * sizeof(double), s));
    GSFM_HIP_CHECK(hipMemsetAsync(ws->jt.get(), 0, ((size_t)(10 + F_) * Mp_ + 64) * sizeof(double2), s));
    hipLaunchKernelGGL(k_ba_obs_ik, dim3(grid_for(M_, kBlock)), dim3(kBlock), 0, s, M_, ws->cam.get(),
                       ws->cam_intr.get(), ws->obs_ik.ensure(M_ + 1));
    ws->qn.ensure(4 * (size_t)Np_);
    ws->tn.ensure(3 * (size_t)Np_);
    ws->camR.ensure(9 * (size_t)Np_);
    ws->camRn.ensure(9 * (size_t)Np_);
    ws->Xn.ensure(3 * (size_t)P_ + 3);
    ws->parn.ensure(KP * (size_t)K_);
    ws->ptH.ensure(9 * (size_t)P_ + 9);
    ws->ptb.ensure(kPtbBa * (size_t)P_ + kPtbBa);
    ws->pth.ensure(6 * (size_t)P_ + 6);
    ws->ptrec.ensure(8 * (size_t)P_ + 8);
    for (DevBuf<double>* b : {&ws->ptdiag, &ws->ptjs}) b->ensure(3 * (size_t)P_ + 3);
    for (DevBuf<double>* b : {&ws->diag, &ws->js, &ws->dvec, &ws->grad, &ws->gred, &ws->rhs, &ws->cg_x, &ws->cg_r,
                              &ws->cg_z, &ws->cg_p, &ws->cg_s})
      b->ensure(n_);
    ws->cg_w.ensure((size_t)n_ + 2);
    ws->spose.ensure(21 * (size_t)Np_);
    if (joint_) {
      ws->scross.ensure(48 * (size_t)N_);
      ws->minvj.ensure(196 * (size_t)N_);
      ws->zrec.ensure((size_t)(6 + F_) * N_ + 2);
      GSFM_HIP_CHECK(hipMemsetAsync(ws->zrec.get(), 0, ((size_t)(6 + F_) * N_ + 2) * sizeof(double), s));
    }
    ws->ipart.ensure(kIntrAcc * (size_t)NI_);
    ws->yi_part.ensure(KP * (size_t)NI_);
    if (rig_) {
      // image tables: frame of each image, its constant cam_from_rig as (R row-major | t), the images of the constant
      // frame, and the frame -> images lists (ascending image order: a fixed summation order)
      std::vector<double> h_cfr, h_sens(12 * (size_t)NI_);
      to_host(ctx_, h_cfr, prob->image_cam_from_rig, 7 * (size_t)NI_, mem);
      const double* h_sen = prob->sensor_cam_from_rig;  // host by contract
      std::vector<unsigned char> h_fix((size_t)NI_, 0);
      std::vector<int> foff((size_t)N_ + 1, 0), fimg((size_t)NI_);
      for (int i = 0; i < NI_; ++i) {
        // the cam_from_rig of an image with a sensor block is the block's value (start value when it is optimised)
        const double* qv = (num_sensors > 0 && h_ims[i] >= 0) ? h_sen + 7 * (size_t)h_ims[i] : &h_cfr[7 * (size_t)i];
        const double w = qv[0], x = qv[1], y = qv[2], z = qv[3];
        double* S = &h_sens[12 * (size_t)i];
        S[0] = 1 - 2 * (y * y + z * z); S[1] = 2 * (x * y - w * z); S[2] = 2 * (x * z + w * y);
        S[3] = 2 * (x * y + w * z); S[4] = 1 - 2 * (x * x + z * z); S[5] = 2 * (y * z - w * x);
        S[6] = 2 * (x * z - w * y); S[7] = 2 * (y * z + w * x); S[8] = 1 - 2 * (x * x + y * y);
        S[9] = qv[4]; S[10] = qv[5]; S[11] = qv[6];
        h_fix[i] = (!sens_ && h_imf[i] == prob->fixed_cam) ? 1 : 0;  // sensor mode: the masks act on the frame rows
        foff[h_imf[i] + 1]++;
      }
      for (int f = 0; f < N_; ++f) foff[f + 1] += foff[f];
      std::vector<int> cur(foff.begin(), foff.end() - 1);
      for (int i = 0; i < NI_; ++i) fimg[cur[h_imf[i]]++] = i;
      GSFM_HIP_CHECK(hipMemcpyAsync(ws->img_frame.ensure(NI_), h_imf.data(), (size_t)NI_ * sizeof(int), hipMemcpyHostToDevice, s));
      GSFM_HIP_CHECK(hipMemcpyAsync(ws->foff.ensure(N_ + 1), foff.data(), (size_t)(N_ + 1) * sizeof(int), hipMemcpyHostToDevice, s));
      GSFM_HIP_CHECK(hipMemcpyAsync(ws->fimg.ensure(NI_), fimg.data(), (size_t)NI_ * sizeof(int), hipMemcpyHostToDevice, s));
      GSFM_HIP_CHECK(hipMemcpyAsync(ws->img_fixed.ensure(NI_), h_fix.data(), (size_t)NI_, hipMemcpyHostToDevice, s));
      GSFM_HIP_CHECK(hipMemcpyAsync(ws->sens.ensure(12 * (size_t)NI_), h_sens.data(), 12 * (size_t)NI_ * sizeof(double), hipMemcpyHostToDevice, s));
      rg_ = RigDev{};
      if (sens_) {
        std::vector<int> soff((size_t)S_ + 1, 0), simg;
        for (int i = 0; i < NI_; ++i)
          if (h_ims[i] >= 0) soff[h_ims[i] + 1]++;
        for (int k = 0; k < S_; ++k) soff[k + 1] += soff[k];
        simg.resize((size_t)soff[S_] + 1);
        std::vector<int> scur(soff.begin(), soff.end() - 1);
        for (int i = 0; i < NI_; ++i)
          if (h_ims[i] >= 0) simg[scur[h_ims[i]]++] = i;
        std::vector<unsigned char> h_fm((size_t)N_);
        for (int f = 0; f < N_; ++f) {
          const bool fixed = f == prob->fixed_cam;  // ba.cc:261-266
          h_fm[f] = (unsigned char)((opt_.optimize_rotations && !fixed ? 1 : 0) | (opt_.optimize_translation && !fixed ? 2 : 0));
        }
        std::vector<double> h_q(4 * (size_t)S_), h_t(3 * (size_t)S_);
        for (int k = 0; k < S_; ++k) {
          for (int j = 0; j < 4; ++j) h_q[4 * (size_t)k + j] = h_sen[7 * (size_t)k + j];
          for (int j = 0; j < 3; ++j) h_t[3 * (size_t)k + j] = h_sen[7 * (size_t)k + 4 + j];
        }
        GSFM_HIP_CHECK(hipMemcpyAsync(ws->img_sensor.ensure(NI_), h_ims.data(), (size_t)NI_ * sizeof(int), hipMemcpyHostToDevice, s));
        GSFM_HIP_CHECK(hipMemcpyAsync(ws->soff.ensure(S_ + 1), soff.data(), (size_t)(S_ + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        GSFM_HIP_CHECK(hipMemcpyAsync(ws->simg.ensure(simg.size()), simg.data(), simg.size() * sizeof(int), hipMemcpyHostToDevice, s));
        GSFM_HIP_CHECK(hipMemcpyAsync(ws->fmask.ensure(N_), h_fm.data(), (size_t)N_, hipMemcpyHostToDevice, s));
        GSFM_HIP_CHECK(hipMemcpyAsync(ws->q.get() + 4 * (size_t)N_, h_q.data(), h_q.size() * sizeof(double), hipMemcpyHostToDevice, s));
        GSFM_HIP_CHECK(hipMemcpyAsync(ws->t.get() + 3 * (size_t)N_, h_t.data(), h_t.size() * sizeof(double), hipMemcpyHostToDevice, s));
        ws->lever.ensure(3 * (size_t)NI_);
        ws->gram_i.ensure(21 * (size_t)NI_);
        GSFM_HIP_CHECK(hipMemsetAsync(ws->gram_i.get(), 0, 21 * (size_t)NI_ * sizeof(double), s));
        rg_.img_sensor = ws->img_sensor.get();
        rg_.lever = ws->lever.get();
        rg_.soff = ws->soff.get();
        rg_.simg = ws->simg.get();
        rg_.fmask = ws->fmask.get();
      }
      GSFM_HIP_CHECK(hipStreamSynchronize(s));  // the host vectors above go out of scope
      rg_.NI = NI_;
      rg_.N = N_;
      rg_.S = S_;
      rg_.img_frame = ws->img_frame.get();
      rg_.sens = ws->sens.get();
      rg_.foff = ws->foff.get();
      rg_.fimg = ws->fimg.get();
      for (DevBuf<double>* b : {&ws->Ri, &ws->Rin}) b->ensure(9 * (size_t)NI_);
      for (DevBuf<double>* b : {&ws->ti, &ws->tin}) b->ensure(3 * (size_t)NI_);
      for (DevBuf<double>* b : {&ws->diag_i, &ws->grad_i, &ws->dvec_i, &ws->zimg, &ws->ximg}) b->ensure((size_t)ni_);
      ws->wimg.ensure((size_t)ni_ + 2);
      ws->gred_i.ensure(6 * (size_t)NI_);
      ws->spose_i.ensure(21 * (size_t)NI_);
      GSFM_HIP_CHECK(hipMemsetAsync(ws->diag_i.get(), 0, (size_t)ni_ * sizeof(double), s));
      GSFM_HIP_CHECK(hipMemsetAsync(ws->grad_i.get(), 0, (size_t)ni_ * sizeof(double), s));
    }
    ws->iacc16.ensure(2 * KP * (size_t)K_);
    ws->iacc44.ensure(kIntrAcc * (size_t)K_);
    ws->minv.ensure(36 * (size_t)Np_ + KP * KP * (size_t)K_);
    ws->vpart.ensure(2 * kCgMaxBlocks * 2);
    ws->dpart.ensure(2 * kMaxApplySlots);
    ws->part.ensure(kMaxBlocks * 8);
    ws->scal.ensure(64);
    ws->cgst.ensure(1);
    ws->cgsc.ensure(2);
    gridP_ = grid_for(P_, kBlock);
    gridN_ = grid_for(N_, kBlock);
    gridNp_ = grid_for(Np_, kBlock);
    gridNI_ = grid_for(NI_, kBlock);
    gridM_ = grid_for(M_, kBlock);
    gridCam_ = grid_wide(g_.g.S, kBlock / 64, kMaxApplySlots);  // one wave per camera segment (delta partial per block)
    gridMulti_ = g_.g.nmulti > 0 ? grid_for(g_.g.nmulti, kBlock / 64) : 0;  // combine pass: one wave per cut camera
    gridTile_ = grid_wide(g_.g.T, kBlock / 64);             // one wave per tile
    gridK_ = small_groups_ ? grid_for(K_, kBlock) : grid_for(K_, 1);
    gridTileP_ = grid_wide(g_.g.T, kBlock / 64, kMaxBlocks);  // tile sweeps that write per-block partials
    g_.K = K_;
    g_.F = F_;
    g_.Mp = Mp_;
    g_.xy = ws->xy.get();
    g_.c_xy = ws->c_xy.get();
    g_.cam_intr = ws->cam_intr.get();
    g_.intr_model = ws->intr_model.get();
    g_.intr_free = ws->intr_free.get();
    g_.intr_map = ws->intr_map.get();
    g_.ioff = ws->ioff.get();
    g_.icams = ws->icams.get();
    g_.obs_ik = ws->obs_ik.get();
    g_.zrec = joint_ ? ws->zrec.get() : nullptr;
    g_.fixed_cam = rig_ ? -1 : prob->fixed_cam;
    g_.img_fixed = (rig_ && !sens_) ? ws->img_fixed.get() : nullptr;
    // sensor mode: image Jacobians unmasked, the flags act on the frame rows (RigDev::fmask)
    g_.opt_rot = (sens_ || opt_.optimize_rotations) ? 1 : 0;
    g_.opt_trn = (sens_ || opt_.optimize_translation) ? 1 : 0;
    g_.opt_pts = opt_.optimize_points ? 1 : 0;
    g_.huber_a = opt_.thres_loss_function;
    g_.lm_lo = opt_.lm.min_lm_diagonal;
    g_.lm_hi = opt_.lm.max_lm_diagonal;
    g1_ = g_;
    g1_.g.pass = 1;  // device view of the combine pass of the camera-major kernels (obsgraph.hpp)
    aw_closed_ok_ = true;
    nfix_ = 0;
    if (g_.fixed_cam >= 0 && g_.fixed_cam < N_ && !rig_) {  // k_ba_aw_modes keeps ONE slot per point for the constant camera's share
      const std::vector<int> slots = cam_slots(ws->og, g_.fixed_cam);  // the constant camera's camera-major slots
      nfix_ = (int)slots.size();
      fix_slots_ = ws->fix_slots.ensure(slots.size() + 1);
      if (nfix_ > 0) {
        GSFM_HIP_CHECK(hipMemcpyAsync(ws->fix_slots.get(), slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice, s));
        GSFM_HIP_CHECK(hipStreamSynchronize(s));
      }
      if (nfix_ > 1) {
        int* flag = ws->og.flag.ensure(4);
        GSFM_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_ba_dup_check, dim3(grid_for((size_t)nfix_, kBlock)), dim3(kBlock), 0, s, g_.g.c_pt, (const int*)fix_slots_,
                           nfix_, flag);
        int* h = reinterpret_cast<int*>(ctx_->h_pinned + 512);
        GSFM_HIP_CHECK(hipMemcpyAsync(h, flag, sizeof(int), hipMemcpyDeviceToHost, s));
        GSFM_HIP_CHECK(hipStreamSynchronize(s));
        aw_closed_ok_ = h[0] == 0;
      }
    }
    q_ = ws->q.get(); qn_ = ws->qn.get();
    t_ = ws->t.get(); tn_ = ws->tn.get();
    R_ = ws->camR.get(); Rn_ = ws->camRn.get();
    X_ = ws->X.get(); Xn_ = ws->Xn.get();
    par_ = ws->par.get(); parn_ = ws->parn.get();
    hipLaunchKernelGGL(k_ba_cam_prepare, dim3(gridNp_), dim3(kBlock), 0, s, Np_, q_, R_);
    // what the sweeps see as camera poses: the frames' own, or the images' cam_from_rig * rig_from_world
    Rk_ = rig_ ? ws->Ri.get() : R_;
    Rkn_ = rig_ ? ws->Rin.get() : Rn_;
    tk_ = rig_ ? ws->ti.get() : t_;
    tkn_ = rig_ ? ws->tin.get() : tn_;
    if (rig_) image_poses(R_, t_, Rk_, tk_);
    // tracks without observations are never visited by the lane-per-observation sweeps: both point
    // buffers start equal, so such tracks keep their input xyz whichever buffer ends up current
    GSFM_HIP_CHECK(hipMemcpyAsync(Xn_, X_, 3 * (size_t)P_ * sizeof(double), hipMemcpyDeviceToDevice, s));
    cg_.n = n_;
    cg_.N = Np_;
    cg_.K = K_;
    cg_.nb_update = joint_ ? std::min(kCgUpdateBlocks, grid_for(N_, kJointCams)) : std::min(kCgUpdateBlocks, grid_for(Np_ + K_, kBlock));
    cg_.zrec = joint_ ? ws->zrec.get() : nullptr;
    cg_.zrec_stride = 6 + F_;
    cg_.zrec_slot = joint_ ? ws->intr_slot.get() : nullptr;
    cg_.joint_map = joint_ ? ws->cam_intr.get() : nullptr;
    cg_.minv_joint = joint_ ? ws->minvj.get() : nullptr;
    cg_.nb_apply = gridCam_ + gridK_ + gridMulti_ + (rig_ ? gridN_ + S_ : 0);  // per-block | phase-I | combine pass | rig damping shares
    cg_.b = ws->rhs.get();
    cg_.x = ws->cg_x.get();
    cg_.r = ws->cg_r.get();
    cg_.z = ws->cg_z.get();
    cg_.p = ws->cg_p.get();
    cg_.s = ws->cg_s.get();
    cg_.w = ws->cg_w.get();
    cg_.minv = ws->minv.get();
    cg_.vpart = ws->vpart.get();
    cg_.dpart = ws->dpart.ensure(std::max((size_t)2 * kMaxApplySlots, (size_t)cg_.nb_apply + 8));
    cg_.scal = ws->cgsc.get();
    cg_.st = ws->cgst.get();
  }

  long used_observations() const { return m_used_; }

  // image poses = cam_from_rig * rig_from_world
  void image_poses(const double* Rp, const double* tp, double* Ri, double* ti) {
    hipLaunchKernelGGL(k_ba_rig_poses, dim3(gridNI_), dim3(kBlock), 0, ctx_->stream, rg_, Rp, tp, Ri, ti);
  }

  double linearize(double* grad_max_norm) override {
    BaWs* ws = ws_;
    hipStream_t s = ctx_->stream;
    double* diag_k = rig_ ? ws->diag_i.get() : ws->diag.get();  // per graph camera (image); summed per frame below
    double* grad_k = rig_ ? ws->grad_i.get() : ws->grad.get();
    const double* sens = rig_ ? ws->sens.get() : nullptr;
    if (sens_) {  // the images' cam_from_rig and levers at this linearisation point
      hipLaunchKernelGGL(k_ba_rig_refresh, dim3(gridNI_), dim3(kBlock), 0, s, rg_, R_, t_);
      diag_k = ws->gram_i.get();
    }
    dispatch_f(F_, [&](auto Fc) {
      WIDE_LAUNCH((k_ba_lin_track<decltype(Fc)::value, WIDE>), dim3(gridTileP_), dim3(kBlock), 0, s, g_, Rk_, tk_, X_, par_,
                         ws->jt.get(), ws->ptdiag.get(), ws->ptH.get(), ws->part.get());
    });
    if (sens_) {
      WIDE_LAUNCH((k_ba_lin_cam<true, WIDE>), dim3(gridCam_), dim3(kBlock), 0, s, g_, Rk_, tk_, X_, par_, ws->c_w.get(),
                         diag_k, grad_k, ws->ipart.get(), sens);
      if (gridMulti_)
        WIDE_LAUNCH((k_ba_lin_cam<true, WIDE>), dim3(gridMulti_), dim3(kBlock), 0, s, g1_, Rk_, tk_, X_, par_,
                           ws->c_w.get(), diag_k, grad_k, ws->ipart.get(), sens);
      hipLaunchKernelGGL(k_ba_rig_frame_lin, dim3(gridN_), dim3(kBlock), 0, s, rg_, diag_k, grad_k, ws->diag.get(),
                         ws->grad.get());
      hipLaunchKernelGGL(k_ba_rig_sensor_lin, dim3(S_), dim3(kBlock), 0, s, rg_, diag_k, grad_k, ws->diag.get(),
                         ws->grad.get());
    } else {
      WIDE_LAUNCH((k_ba_lin_cam<false, WIDE>), dim3(gridCam_), dim3(kBlock), 0, s, g_, Rk_, tk_, X_, par_, ws->c_w.get(),
                         diag_k, grad_k, ws->ipart.get(), sens);
      if (gridMulti_)  // combine pass over the cameras whose lists were cut into slices
        WIDE_LAUNCH((k_ba_lin_cam<false, WIDE>), dim3(gridMulti_), dim3(kBlock), 0, s, g1_, Rk_, tk_, X_, par_,
                           ws->c_w.get(), diag_k, grad_k, ws->ipart.get(), sens);
    }
    if (rig_ && !sens_) {  // lin_cam accumulated in the frame tangent: the frame's values are plain sums over its images
      hipLaunchKernelGGL((k_ba_rig_sum<6>), dim3(gridN_), dim3(kBlock), 0, s, N_, ws->foff.get(), ws->fimg.get(), diag_k,
                         ws->diag.get());
      hipLaunchKernelGGL((k_ba_rig_sum<6>), dim3(gridN_), dim3(kBlock), 0, s, N_, ws->foff.get(), ws->fimg.get(), grad_k,
                         ws->grad.get());
    }
    group_sum<2 * KP>(ws->ipart.get(), ws->iacc16.get());
    hipLaunchKernelGGL(k_ba_intr_unpack16, dim3(grid_for(KP * (size_t)K_, kBlock)), dim3(kBlock), 0, s, Np_, K_,
                       ws->iacc16.get(), ws->diag.get(), ws->grad.get());
    if (ctx_->comm.world > 1) {
      allreduce_sum(ctx_, ws->diag.get(), n_);
      allreduce_sum(ctx_, ws->grad.get(), n_);
    }
    const int gmx = std::min(64, grid_for((size_t)n_, kBlock));
    hipLaunchKernelGGL(k_ba_absmax, dim3(gmx), dim3(kBlock), 0, s, (const double*)ws->grad.get(), n_, ws->maxpart.ensure(64));
    hipLaunchKernelGGL(k_ba_finalize_lin, dim3(1), dim3(kBlock), 0, s, ws->part.get(), gridTileP_, (const double*)ws->maxpart.get(), gmx,
                       ws->scal.get());
    if (ctx_->comm.world > 1) {
      allreduce_sum(ctx_, ws->scal.get(), 1);
      allreduce_max(ctx_, ws->scal.get() + 1, 1);
    }
    GSFM_HIP_CHECK(hipMemcpyAsync(ctx_->h_pinned + 300, ws->scal.get(), 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    GSFM_HIP_CHECK(hipGetLastError());
    comm_check(ctx_);
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
    hipLaunchKernelGGL(k_ba_build_track, dim3(gridP_), dim3(kBlock), 0, s, g_, radius, X_, ws->ptH.get(),
                       ws->ptdiag.get(), ws->ptjs.get(), ws->ptb.get(), ws->ptrec.get(), ws->pth.get());
    if constexpr (KP == 8) {  // (the joint layout exists in the 8-wide unit only: joint_ is false otherwise)
     if (joint_) {
      dispatch_f(F_, [&](auto Fc) {
        constexpr int F = decltype(Fc)::value;
        WIDE_LAUNCH((k_ba_build_cam<true, WIDE, F>), dim3(gridCam_), dim3(kBlock), 0, s, g_, R_, t_, par_, ws->c_w.get(),
                    ws->ptb.get(), ws->gred.get(), ws->spose.get(), ws->ipart.get(), ws->scross.get());
        if (gridMulti_)
          WIDE_LAUNCH((k_ba_build_cam<true, WIDE, F>), dim3(gridMulti_), dim3(kBlock), 0, s, g1_, R_, t_, par_, ws->c_w.get(),
                      ws->ptb.get(), ws->gred.get(), ws->spose.get(), ws->ipart.get(), ws->scross.get());
      });
     }
    }
    if (!joint_) {
      double* gred_k = rig_ ? ws->gred_i.get() : ws->gred.get();
      double* spose_k = rig_ ? ws->spose_i.get() : ws->spose.get();
      dispatch_f(F_, [&](auto Fc) {
        constexpr int F = decltype(Fc)::value;
        WIDE_LAUNCH((k_ba_build_cam<false, WIDE, F>), dim3(gridCam_), dim3(kBlock), 0, s, g_, Rk_, tk_, par_, ws->c_w.get(),
                    ws->ptb.get(), gred_k, spose_k, ws->ipart.get(), (double*)nullptr);
        if (gridMulti_)
          WIDE_LAUNCH((k_ba_build_cam<false, WIDE, F>), dim3(gridMulti_), dim3(kBlock), 0, s, g1_, Rk_, tk_, par_, ws->c_w.get(),
                      ws->ptb.get(), gred_k, spose_k, ws->ipart.get(), (double*)nullptr);
      });
      if (rig_)  // frame blocks: sum over the frame's images of T^T g and T^T S T (cross blocks between two images of
                 // one frame are left to the PCG: this is the preconditioner and the right-hand side)
        hipLaunchKernelGGL(k_ba_rig_reduce_blocks, dim3(gridN_), dim3(kBlock), 0, s, rg_, gred_k, spose_k, ws->gred.get(),
                           ws->spose.get());
      if (sens_)
        hipLaunchKernelGGL(k_ba_rig_sensor_blocks, dim3(S_), dim3(kBlock), 0, s, rg_, gred_k, spose_k, ws->gred.get(),
                           ws->spose.get());
    }
    group_sum<kIntrAcc>(ws->ipart.get(), ws->iacc44.get());
    if (multi) {
      allreduce_sum(ctx_, ws->gred.get(), 6 * (size_t)Np_);
      allreduce_sum(ctx_, ws->spose.get(), 21 * (size_t)Np_);
      allreduce_sum(ctx_, ws->iacc44.get(), kIntrAcc * (size_t)K_);
      if (joint_) allreduce_sum(ctx_, ws->scross.get(), 48 * (size_t)N_);
    }
    if (joint_) {
      hipLaunchKernelGGL(k_ba_blocks_finalize_joint, dim3(grid_for(N_, kBlock / 16)), dim3(kBlock), 0, s, N_, radius, g_.lm_lo,
                         g_.lm_hi, g_.cam_intr, ws->diag.get(), ws->js.get(), ws->gred.get(), ws->spose.get(),
                         ws->iacc44.get(), ws->scross.get(), ws->dvec.get(), ws->rhs.get(), ws->minvj.get());
    } else
    hipLaunchKernelGGL(k_ba_blocks_finalize, dim3(grid_for(Np_ + K_, kBlock)), dim3(kBlock), 0, s, Np_, K_, radius,
                       g_.lm_lo, g_.lm_hi, ws->diag.get(), ws->js.get(), ws->gred.get(), ws->spose.get(),
                       ws->iacc44.get(), ws->dvec.get(), ws->rhs.get(), ws->minv.get());
    if (rig_)
      hipLaunchKernelGGL(k_ba_rig_dvec, dim3(grid_for((size_t)ni_, kBlock)), dim3(kBlock), 0, s, NI_, Np_, K_, ws->dvec.get(),
                         ws->dvec_i.get());
    *linear_iterations = pcg();
    const double* dy_k = ws->cg_x.get();
    if (rig_) {  // the step of an image's pose is T_s times the step of its frame
      hipLaunchKernelGGL(k_ba_rig_expand, dim3(grid_for((size_t)NI_ + K_, kBlock)), dim3(kBlock), 0, s, rg_, Np_, K_,
                         ws->cg_x.get(), ws->ximg.get());
      dy_k = ws->ximg.get();
    }
    dispatch_f(F_, [&](auto Fc) {
      hipLaunchKernelGGL((k_ba_backsub<decltype(Fc)::value>), dim3(gridTileP_), dim3(kBlock), 0, s, g_, X_, ws->jt.get(),
                         ws->ptb.get(), dy_k, Xn_, ws->part.get());
    });
    hipLaunchKernelGGL((k_ba_sum_partials<3>), dim3(1), dim3(kBlock), 0, s, ws->part.get(), gridTileP_, ws->scal.get());
    const int gridU = std::min(64, grid_for(Np_ + KP * (size_t)K_, kBlock));
    double* part2 = ws->part.get() + kMaxBlocks * 3;
    hipLaunchKernelGGL(k_ba_param_update, dim3(gridU), dim3(kBlock), 0, s, Np_, K_, q_, t_, par_, ws->cg_x.get(), qn_,
                       tn_, parn_, part2);
    hipLaunchKernelGGL((k_ba_sum_partials<3>), dim3(1), dim3(kBlock), 0, s, part2, gridU, ws->scal.get() + 3);
    hipLaunchKernelGGL(k_ba_cam_prepare, dim3(gridNp_), dim3(kBlock), 0, s, Np_, qn_, Rn_);
    if (rig_) image_poses(Rn_, tn_, Rkn_, tkn_);
    double* part3 = ws->part.get() + kMaxBlocks * 4;
    WIDE_LAUNCH((k_ba_cost<WIDE>), dim3(gridM_), dim3(kBlock), 0, s, g_, Rkn_, tkn_, Xn_, parn_, part3);
    hipLaunchKernelGGL((k_ba_sum_partials<1>), dim3(1), dim3(kBlock), 0, s, part3, gridM_, ws->scal.get() + 6);
    if (multi) {
      allreduce_sum(ctx_, ws->scal.get(), 3);
      allreduce_sum(ctx_, ws->scal.get() + 6, 1);
    }
    double h[7];
    GSFM_HIP_CHECK(hipMemcpyAsync(ctx_->h_pinned + 256, ws->scal.get(), 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    GSFM_HIP_CHECK(hipGetLastError());
    comm_check(ctx_);
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
    std::swap(Rk_, Rkn_);  // (trivial rigs: the same buffers as R_ / t_)
    std::swap(tk_, tkn_);
    std::swap(X_, Xn_);
    std::swap(par_, parn_);
  }

  void write_back(const gsfm_ba_problem* prob, double* cam_q, double* cam_t, double* pt_xyz, double* intr) {
    copy_out(ctx_, cam_q, q_, 4 * (size_t)N_, prob->mem);
    copy_out(ctx_, cam_t, t_, 3 * (size_t)N_, prob->mem);
    copy_out(ctx_, pt_xyz, X_, 3 * (size_t)P_, prob->mem);
    copy_out(ctx_, intr, par_, KP * (size_t)K_, prob->mem);
    std::vector<double> h_q(4 * (size_t)S_), h_t(3 * (size_t)S_);
    if (sens_) {
      GSFM_HIP_CHECK(hipMemcpyAsync(h_q.data(), q_ + 4 * (size_t)N_, h_q.size() * sizeof(double), hipMemcpyDeviceToHost, ctx_->stream));
      GSFM_HIP_CHECK(hipMemcpyAsync(h_t.data(), t_ + 3 * (size_t)N_, h_t.size() * sizeof(double), hipMemcpyDeviceToHost, ctx_->stream));
    }
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx_->stream));
    for (int k = 0; k < S_; ++k) {  // the optimised cam_from_rig blocks, in place (host table)
      for (int j = 0; j < 4; ++j) prob->sensor_cam_from_rig[7 * (size_t)k + j] = h_q[4 * (size_t)k + j];
      for (int j = 0; j < 3; ++j) prob->sensor_cam_from_rig[7 * (size_t)k + 4 + j] = h_t[3 * (size_t)k + j];
    }
  }

 private:
  template <int W>
  void group_sum(const double* part, double* out) {
    hipStream_t s = ctx_->stream;
    if (small_groups_) {
      hipLaunchKernelGGL((k_ba_group_sum_small<W>), dim3(gridK_), dim3(kBlock), 0, s, K_, g_.ioff, g_.icams, part, out);
    } else {
      hipLaunchKernelGGL((k_ba_group_sum<W>), dim3(gridK_), dim3(kBlock), 0, s, K_, g_.ioff, g_.icams, part, out);
    }
  }

  // (S + D) x = rhs by a dense inverse (k_ba_dense_assemble / _finish, dense_spd_solve of ra_dense.hpp): into cg_x
  bool dense_solve() {  // false: the dense inverse did not reach the tolerance (cg_x is then not a solution)
    if constexpr (KP == 8) {
      BaWs* ws = ws_;
      hipStream_t s = ctx_->stream;
      const int ld = (n_ + kTile - 1) / kTile * kTile;
      const size_t nn = (size_t)ld * ld;
      double* S0 = ws->dn_S.ensure(nn);
      double* bufA = ws->dn_a.ensure(nn);
      double* bufB = ws->dn_b.ensure(nn);
      double* pinv = ws->dn_pinv.ensure((size_t)(ld / kTile) * kTile * kTile);
      double* r = ws->dn_r.ensure(ld);
      double* dx = ws->dn_dx.ensure(ld);
    double* sc = ws->dn_sc.ensure(ld);
      const int cw = std::min(ld, kBaDenseWindow), nwin = (ld + cw - 1) / cw;
      const size_t lds = (6 * (size_t)cw + (size_t)KP * KP * K_) * sizeof(double);
      if (lds > 64 * 1024)
        GSFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_dense_assemble), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GSFM_HIP_CHECK(hipMemsetAsync(S0, 0, nn * sizeof(double), s));
      hipLaunchKernelGGL(k_ba_dense_assemble, dim3(N_, nwin), dim3(kBlock), lds, s, g_, (const double2*)ws->jt.get(), (const double*)ws->pth.get(), ld, cw, S0);
      hipLaunchKernelGGL(k_ba_dense_finish, dim3(grid_wide(nn, kBlock, 1 << 12)), dim3(kBlock), 0, s, N_, n_, ld, (const double*)ws->dvec.get(), S0);
      if (const char* dd = std::getenv("GSFM_DUMP_DENSE")) {  // diagnostics (tools/exp_capture_ba_verbose.py): the assembled matrix and rhs
        static int counter = 0;
        std::vector<double> h(nn + (size_t)ld);
        GSFM_HIP_CHECK(hipMemcpyAsync(h.data(), S0, nn * sizeof(double), hipMemcpyDeviceToHost, s));
        GSFM_HIP_CHECK(hipMemcpyAsync(h.data() + nn, ws->rhs.get(), (size_t)n_ * sizeof(double), hipMemcpyDeviceToHost, s));
        GSFM_HIP_CHECK(hipStreamSynchronize(s));
        char path[512];
        snprintf(path, sizeof path, "%s/dense_%d_%d_%d.bin", dd, n_, ld, counter++);
        if (FILE* f = fopen(path, "wb")) {
          fwrite(h.data(), sizeof(double), h.size(), f);
          fclose(f);
        }
      }
      const bool ok = dense_spd_solve(s, n_, ld, S0, bufA, bufB, pinv, r, dx, sc, ws->dn_nrm.ensure(2), (const double*)ws->rhs.get(), ws->cg_x.get(),
                                      opt_.lm.pcg_relative_tolerance);
      if (ok) ctx_->stats[GSFM_STAT_DENSE_SOLVES]++;
      return ok;
    }
    return false;
  }
  bool dense_ok() const {
    return KP == 8 && !rig_ && ctx_->comm.world == 1 && n_ <= kBaDenseMaxUnknowns && K_ <= kBaDenseMaxIntr &&
           ctx_->knob[GSFM_KNOB_GP_DENSE] != 1;
  }

  long pcg() {
    BaWs* ws = ws_;
    hipStream_t s = ctx_->stream;
    bool dense_failed = false;  // this system is beyond the dense inverse (dense_spd_solve): PCG, uncapped
    if (dense_ok() && (dense_on_ || ctx_->knob[GSFM_KNOB_GP_DENSE] == 2)) {
      if (dense_solve()) return 0;
      dense_failed = true;
    }
    const bool may_dense = !dense_failed && dense_ok() && opt_.lm.pcg_max_iterations > kBaDenseTrigger;
    bool finished = false;  // (cg_solve only ever sets it)
    const double yscale = ctx_->comm.rank == 0 ? 1.0 : 0.0;
    const double tol = opt_.lm.pcg_relative_tolerance;
    // the similarity gauge of the scene deflated from the PCG (CgDeflation, cg.hpp): trivial rigs, translations among
    // the unknowns; skipped while the solves are short anyway (strongly damped LM steps; defl_on_, below)
    CgDeflation defl;
    // (the closed-form A W sweep exists in the 8-wide unit only; the 16-wide unit forms the products by operator applications)
    if (!rig_ && g_.opt_trn && defl_on_ && N_ >= 64) {
      const int with_rot = g_.opt_rot ? 1 : 0;
      defl.k = with_rot ? 7 : 4;
      const size_t n = (size_t)n_;
      double* W = ws->defl_w.ensure(defl.k * n);
      defl.AW = ws->defl_aw.ensure(defl.k * n);
      defl.b2 = ws->defl_b2.ensure(n);
      defl.part = ws->defl_part.ensure((size_t)kCgdBlocks * kCgdGram);
      defl.small = ws->defl_small.ensure(80);
      defl.cd = ws->defl_cd.ensure((size_t)2 * kCgMaxBlocks * 2 * kCgMaxModes);
      GSFM_HIP_CHECK(hipMemsetAsync(W, 0, defl.k * n * sizeof(double), s));
      hipLaunchKernelGGL(k_ba_defl_modes, dim3(gridN_), dim3(kBlock), 0, s, N_, (long)n_, (const double*)Rk_, (const double*)tk_,
                         g_.fixed_cam, with_rot, W);
      defl.W = W;
      // A W in closed form (k_ba_aw_modes): points optimised, no point observed twice by the constant camera
      const bool no_closed = KP > 8 || ctx_->knob[GSFM_KNOB_BA_AW_BY_APPLICATION] != 0;  // A/B and tests: form A W by operator applications
      if constexpr (KP == 8)
      if (!no_closed && g_.opt_pts && aw_closed_ok_) {
        // intrinsics blocks shared by several cameras (not the joint layout): per-camera shares, then per-block sums
        const bool shared_blocks = !joint_ && F_ > 0;
        double* ipart = shared_blocks ? ws->awi_part.ensure((size_t)N_ * defl.k * KP) : nullptr;
        GSFM_HIP_CHECK(hipMemsetAsync(defl.AW, 0, defl.k * n * sizeof(double), s));
        double* ftab = ws->ftab.ensure(6 * (size_t)std::max(1, nfix_));
        if (nfix_ > 0)
          WIDE_LAUNCH((k_ba_fixed_share<WIDE>), dim3(grid_for((size_t)nfix_, kBlock)), dim3(kBlock), 0, s, g_, (const int*)fix_slots_, nfix_,
                      Rk_, tk_, par_, ws->c_w.get(), ws->ptb.get(), ftab);
        dispatch_f(F_, [&](auto Fc) {
          constexpr int F = decltype(Fc)::value;
          auto launch = [&](auto rot, const BaDev& gd, int grid) {
            constexpr bool ROT = decltype(rot)::value;
            WIDE_LAUNCH((k_ba_aw_modes<ROT, WIDE, F>), dim3(grid), dim3(kBlock), 0, s, gd, yscale, Rk_, tk_, par_, ws->c_w.get(),
                        (const double*)ws->ptb.get(), (const double*)ftab, (const double*)ws->dvec.get(), (const double*)W,
                        defl.AW, (long)n_, ipart);
          };
          if (with_rot) {
            launch(std::true_type{}, g_, gridCam_);
            if (gridMulti_) launch(std::true_type{}, g1_, gridMulti_);
          } else {
            launch(std::false_type{}, g_, gridCam_);
            if (gridMulti_) launch(std::false_type{}, g1_, gridMulti_);
          }
        });
        if (shared_blocks) {
          double* gsum = ws->awi_sum.ensure((size_t)K_ * defl.k * KP);
          if (with_rot) group_sum<7 * KP>(ipart, gsum); else group_sum<4 * KP>(ipart, gsum);
          hipLaunchKernelGGL(k_ba_aw_intr_scatter, dim3(grid_for((size_t)K_ * defl.k * KP, kBlock)), dim3(kBlock), 0, s, N_, K_, defl.k,
                             (const double*)gsum, defl.AW, (long)n_);
        }
        if (ctx_->comm.world > 1) allreduce_sum(ctx_, defl.AW, defl.k * n);
        defl.aw_ready = defl.k;
        // diagnostics: keep the closed-form products, let cg_solve form them by operator applications as well, compare below
        if (ctx_->knob[GSFM_KNOB_BA_AW_CHECK]) {
          aw_check_.resize(defl.k * n);
          GSFM_HIP_CHECK(hipMemcpyAsync(aw_check_.data(), defl.AW, defl.k * n * sizeof(double), hipMemcpyDeviceToHost, s));
          GSFM_HIP_CHECK(hipStreamSynchronize(s));
          defl.aw_ready = 0;
        }
      }
    }
    const long iters = cg_solve<6, true, KP>(ctx_, cg_, tol, may_dense ? kBaDenseTrigger : opt_.lm.pcg_max_iterations, [&](int it) {
      // rigs: the sweeps run on per-image vectors (z_image = T_s z_frame) with a zero pose diagonal; everything else of
      // the PCG state (partials, status, scalars — set by cg_solve on cg_) is shared with the frame-space solve
      CgVec vk = cg_;
      const double* dk = ws->dvec.get();
      if (rig_) {
        hipLaunchKernelGGL(k_ba_rig_expand, dim3(grid_for((size_t)NI_ + K_, kBlock)), dim3(kBlock), 0, s, rg_, Np_, K_, cg_.z,
                           ws->zimg.get());
        vk.z = ws->zimg.get();
        vk.w = ws->wimg.get();
        dk = ws->dvec_i.get();
      }
      bool timed = ctx_->prof.begin(s, GSFM_KERNEL_BA_SCHUR, it);
      dispatch_f(F_, [&](auto Fc) {
        const bool nt = !ctx_->knob[GSFM_KNOB_BA_NO_NONTEMPORAL];  // measured on C4: 240 us -> 210 (loads first) -> 193 (+ non-temporal)
        if (nt)
          hipLaunchKernelGGL((k_ba_phaseA<decltype(Fc)::value, true>), dim3(gridTile_), dim3(kBlock), 0, s, g_, vk, it,
                             tol * tol, ws->jt.get(), ws->pth.get(), ws->ptrec.get());
        else
          hipLaunchKernelGGL((k_ba_phaseA<decltype(Fc)::value, false>), dim3(gridTile_), dim3(kBlock), 0, s, g_, vk, it,
                             tol * tol, ws->jt.get(), ws->pth.get(), ws->ptrec.get());
      });
      if (timed) ctx_->prof.end(s);
      timed = ctx_->prof.begin(s, GSFM_KERNEL_BA_SCHUR_B, it);
      WIDE_LAUNCH((k_ba_phaseB<WIDE>), dim3(gridCam_), dim3(kBlock), 0, s, g_, vk, yscale, Rk_, tk_, par_,
                         ws->c_w.get(), ws->ptrec.get(), dk, ws->yi_part.get(), 0);
      if (gridMulti_)
        WIDE_LAUNCH((k_ba_phaseB<WIDE>), dim3(gridMulti_), dim3(kBlock), 0, s, g1_, vk, yscale, Rk_, tk_, par_,
                           ws->c_w.get(), ws->ptrec.get(), dk, ws->yi_part.get(), gridCam_ + gridK_);
      if (timed) ctx_->prof.end(s);
      if (small_groups_) {
        hipLaunchKernelGGL(k_ba_phaseI_small, dim3(gridK_), dim3(kBlock), 0, s, g_, vk, yscale, ws->yi_part.get(), dk, gridCam_);
      } else {
        hipLaunchKernelGGL(k_ba_phaseI, dim3(gridK_), dim3(kBlock), 0, s, g_, vk, yscale, ws->yi_part.get(), dk, gridCam_);
      }
      if (rig_)
        hipLaunchKernelGGL(k_ba_rig_reduce_w, dim3(gridN_ + S_), dim3(kBlock), 0, s, cg_, rg_, yscale, ws->wimg.get(), ws->dvec.get(),
                           gridCam_ + gridK_ + gridMulti_, gridN_);
    }, defl.k ? &defl : nullptr, &pcg_hint_, [](int) {}, &finished);
    if (may_dense && !finished) {  // still running after kBaDenseTrigger iterations: this and the later solves of the LM problem are direct
      dense_on_ = true;
      if (dense_solve()) return iters;
      return iters + pcg();  // (enters with dense_on_ set: the dense path fails the same way once more and the PCG runs uncapped)
    }
    if (!aw_check_.empty() && defl.k) {
      std::vector<double> applied(aw_check_.size());
      GSFM_HIP_CHECK(hipMemcpyAsync(applied.data(), defl.AW, applied.size() * sizeof(double), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipStreamSynchronize(s));
      for (int j = 0; j < defl.k; ++j) {
        double dmax = 0.0, amax = 0.0;
        long where = -1;
        for (long i = 0; i < n_; ++i) {
          const double a = applied[(size_t)j * n_ + i], d = std::fabs(a - aw_check_[(size_t)j * n_ + i]);
          amax = std::max(amax, std::fabs(a));
          if (d > dmax) dmax = d, where = i;
        }
        fprintf(stderr, "[gsfm ba] A W mode %d: max |closed - applied| = %.3e at %ld (max |applied| = %.3e)\n", j, dmax, where, amax);
      }
      aw_check_.clear();
    }
    // deflation pays while a plain solve needs more than ~3 k iterations (iters includes the applications that formed A W, if any)
    defl_on_ = defl.k ? iters - (defl.k - defl.aw_ready) > defl.k : iters > 3 * 7;
    return iters;
  }

  gsfm_ctx* ctx_;
  BaWs* ws_;
  gsfm_ba_options opt_;
  BaDev g_{}, g1_{};
  CgVec cg_{};
  int N_ = 0, K_ = 0, n_ = 0, F_ = 0, max_group_ = 0;
  int NI_ = 0, ni_ = 0, gridNI_ = 1;  // cameras of the observation graph (= N_, or the images of calibrated rigs)
  int S_ = 0, Np_ = 0, gridNp_ = 1;   // optimised cam_from_rig blocks; pose blocks = frames + sensor blocks
  bool rig_ = false, sens_ = false;
  RigDev rg_{};
  double *Rk_ = nullptr, *Rkn_ = nullptr, *tk_ = nullptr, *tkn_ = nullptr;  // poses the sweeps see (frames or images)
  bool small_groups_ = false, joint_ = false;
  bool wide_ = false;  // some camera uses a fisheye / FOV model: the sweeps run their WIDE instances
  std::vector<double> aw_check_;  // GSFM_BA_AW_CHECK: the closed-form products of the running solve
  int nfix_ = 0;              // observations of the constant camera, their camera-major slots (k_ba_fixed_share)
  int* fix_slots_ = nullptr;
  bool aw_closed_ok_ = true;  // false: a point is observed twice by the constant camera (k_ba_aw_modes keeps one slot per point)
  bool dense_on_ = false;  // the reduced systems of this LM problem are solved densely (dense_solve)
  bool defl_on_ = true;  // deflate the next reduced solve (short solves run plain)
  int pcg_hint_ = 0;     // iteration count of the previous reduced solve (where cg_solve first reads the status back)
  long P_ = 0, M_ = 0, Mp_ = 0, m_used_ = 0;
  int gridP_ = 1, gridN_ = 1, gridM_ = 1, gridCam_ = 1, gridMulti_ = 0, gridTile_ = 1, gridTileP_ = 1, gridK_ = 1;
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
  const int rc = lm_minimize(solver, opt->lm, rep, &ctx->lm_trace);
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
