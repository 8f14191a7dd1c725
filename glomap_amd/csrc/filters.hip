// filters.hip — the per-observation / per-edge processors that run between the estimator calls
// (SURVEY.md section 8f rows 1-2), on MI355X (gfx950).
//
//   TrackFilter::FilterTracksByReprojection     glomap/processors/track_filter.cc:7-52
//   TrackFilter::FilterTracksByAngle            glomap/processors/track_filter.cc:54-90
//   TrackFilter::FilterTrackTriangulationAngle  glomap/processors/track_filter.cc:92-127
//   NormalizeReconstruction                     glomap/processors/reconstruction_normalizer.cc:5-85
//   RelPoseFilter::FilterRotations              glomap/processors/relpose_filter.cc:7-33
//
// They sit between every GP / BA solve of GlobalMapper::Solve (global_mapper.cc:164-186, 231-275,
// 309-333).  All of them are single sweeps: one lane per observation (coalesced 24-byte ray / 16-byte
// pixel reads, L2-resident camera gathers, the track's point read by its own consecutive lanes), or one
// thread per track / edge.  Results are keep masks (bytes) + the counter the reference returns; counters
// are integer sums, so results are bit-reproducible.
#include <algorithm>
#include <cmath>
#include <limits>

#include "camera.hpp"
#include "device.hpp"

namespace gsfm {
// sort.hip (rocPRIM plumbing, instantiated once there)
void exclusive_scan_i64(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const long* in, long* out, size_t n);
namespace {

constexpr double kEps = 1e-12;  // glomap/types.h:14

struct ViewDev {
  int N;
  long P, M;
  const long* off;
  const int* cam;
  const double* undist;
  const double* xy;
  const double* q;
  const double* t;
  const double* X;
  const unsigned char* calibrated;
  const int* cam_intr;
  const int* intr_model;
  const double* intr_params;
};

__device__ __forceinline__ void quat_to_R9(const double* __restrict__ q, double (&R)[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

// point of observation k: binary search of the track in pt_offset (no extra index array needed)
__device__ __forceinline__ long track_of(const long* __restrict__ off, long P, long k) {
  long lo = 0, hi = P;
  while (hi - lo > 1) {
    const long mid = (lo + hi) >> 1;
    if (off[mid] <= k) lo = mid; else hi = mid;
  }
  return lo;
}

// obs_pt[k] = track of observation k (one thread per track writes its run)
__global__ void __launch_bounds__(kBlock)
    k_fill_obs_pt(long P, const long* __restrict__ off, int* __restrict__ obs_pt) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x)
    for (long k = off[p]; k < off[p + 1]; ++k) obs_pt[k] = (int)p;
}

// mode 0: reprojection error in normalised image coordinates; 1: in pixels; 2: angle test.
// One lane per observation: coalesced ray / pixel / index reads, L2-resident camera gathers, the point of
// the track read by its consecutive lanes.
// WIDE: some camera uses a fisheye / FOV model (ids >= OPENCV_FISHEYE); the lean instance leaves their transcendental
// branches out, as the BA kernels do (camera.hpp: with them inlined this sweep fell from 0.35 to 0.28 of the HBM rate).
// KP: doubles per intrinsics row (8, or 16 for the camera models with more than eight parameters: gsfm_scene_view::intr_stride).
template <bool WIDE, int KP>
__global__ void __launch_bounds__(kBlock)
    k_filter_obs(ViewDev v, const int* __restrict__ obs_pt, int mode, double thr, double thr_uncalib,
                 unsigned char* __restrict__ keep) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < v.M; k += (long)gridDim.x * blockDim.x) {
    const long p = obs_pt[k];
    const int n = v.cam[k];
    double R[9];
    quat_to_R9(v.q + 4 * (long)n, R);
    const V3 Xp = ld3(v.X + 3 * p);
    const V3 tc = ld3(v.t + 3 * (long)n);
    const V3 pc{R[0] * Xp.x + R[1] * Xp.y + R[2] * Xp.z + tc.x, R[3] * Xp.x + R[4] * Xp.y + R[5] * Xp.z + tc.y,
                R[6] * Xp.x + R[7] * Xp.y + R[8] * Xp.z + tc.z};
    bool ok = false;
    if (!(pc.z < kEps)) {  // track_filter.cc:21,72
      if (mode == 0) {
        const V3 u = ld3(v.undist + 3 * k);
        const double ex = pc.x / pc.z - u.x / (u.z + kEps), ey = pc.y / pc.z - u.y / (u.z + kEps);
        ok = sqrt(ex * ex + ey * ey) < thr;
      } else if (mode == 1) {
        const int ik = v.cam_intr[n];
        const double iz = 1.0 / pc.z;
        double px = 0.0, py = 0.0, Juv[4], Jp[2][KP];
        // Camera::ImgFromCam(...).value_or(Zero): projection fails for points at / behind the camera
        if (pc.z > 2.220446049250313e-16)
          distort_project<WIDE, KP>(v.intr_model[ik], v.intr_params + KP * (long)ik, pc.x * iz, pc.y * iz, px, py, Juv, Jp);
        const double ex = px - v.xy[2 * k], ey = py - v.xy[2 * k + 1];
        ok = sqrt(ex * ex + ey * ey) < thr;
      } else {
        const V3 u = ld3(v.undist + 3 * k);
        const double inv = 1.0 / sqrt(dot(pc, pc));
        const double c = (pc.x * u.x + pc.y * u.y + pc.z * u.z) * inv;
        const bool cal = v.calibrated == nullptr || v.calibrated[n];
        ok = c > (cal ? thr : thr_uncalib);
      }
    }
    keep[k] = ok ? 1 : 0;
  }
}

// number of tracks that lost at least one observation
__global__ void __launch_bounds__(kBlock)
    k_count_changed(long P, const long* __restrict__ off, const unsigned char* __restrict__ keep,
                    unsigned long long* __restrict__ counter) {
  unsigned long long c = 0;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    bool changed = false;
    for (long k = off[p]; k < off[p + 1]; ++k) changed = changed || keep[k] == 0;
    c += changed ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(counter, c);
}

// Triangulation angle: a track stays iff some pair of its viewing rays (from the camera centres) spans
// more than min_angle.  One thread per track, O(L^2) like the reference (L <= a few dozen).
__global__ void __launch_bounds__(kBlock)
    k_filter_tri(ViewDev v, double cos_thr, unsigned char* __restrict__ keep_track, unsigned long long* __restrict__ counter) {
  unsigned long long c = 0;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < v.P; p += (long)gridDim.x * blockDim.x) {
    const V3 Xp = ld3(v.X + 3 * p);
    const long k0 = v.off[p], k1 = v.off[p + 1];
    bool status = false;
    for (long a = k0; a < k1 && !status; ++a) {
      const int na = v.cam[a];
      double Ra[9];
      quat_to_R9(v.q + 4 * (long)na, Ra);
      const V3 ta = ld3(v.t + 3 * (long)na);
      // Image::Center() = -R^T t  (image.h)
      V3 da = Xp - V3{-(Ra[0] * ta.x + Ra[3] * ta.y + Ra[6] * ta.z), -(Ra[1] * ta.x + Ra[4] * ta.y + Ra[7] * ta.z),
                      -(Ra[2] * ta.x + Ra[5] * ta.y + Ra[8] * ta.z)};
      da = (1.0 / sqrt(dot(da, da))) * da;
      for (long b = a + 1; b < k1; ++b) {
        const int nb = v.cam[b];
        double Rb[9];
        quat_to_R9(v.q + 4 * (long)nb, Rb);
        const V3 tb = ld3(v.t + 3 * (long)nb);
        V3 db = Xp - V3{-(Rb[0] * tb.x + Rb[3] * tb.y + Rb[6] * tb.z), -(Rb[1] * tb.x + Rb[4] * tb.y + Rb[7] * tb.z),
                        -(Rb[2] * tb.x + Rb[5] * tb.y + Rb[8] * tb.z)};
        db = (1.0 / sqrt(dot(db, db))) * db;
        if (dot(da, db) < cos_thr) {
          status = true;
          break;
        }
      }
    }
    keep_track[p] = status ? 1 : 0;
    c += status ? 0 : 1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(counter, c);
}

// camera centres of the registered images as float (reconstruction_normalizer.cc:24-30)
__global__ void __launch_bounds__(kBlock)
    k_centers_f32(int N, const double* __restrict__ q, const double* __restrict__ t, float* __restrict__ c) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    double R[9];
    quat_to_R9(q + 4 * (long)n, R);
    const V3 tt = ld3(t + 3 * (long)n);
    c[3 * (long)n] = (float)(-(R[0] * tt.x + R[3] * tt.y + R[6] * tt.z));
    c[3 * (long)n + 1] = (float)(-(R[1] * tt.x + R[4] * tt.y + R[7] * tt.z));
    c[3 * (long)n + 2] = (float)(-(R[2] * tt.x + R[5] * tt.y + R[8] * tt.z));
  }
}
// TransformCameraWorld(Sim3(scale, I, -scale mean), cam_from_world): t' = scale (t + R mean)
__global__ void __launch_bounds__(kBlock)
    k_transform_cams(int N, double scale, double mx, double my, double mz, const double* __restrict__ q, double* __restrict__ t) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    double R[9];
    quat_to_R9(q + 4 * (long)n, R);
    double* tt = t + 3 * (long)n;
    const double a = tt[0] + R[0] * mx + R[1] * my + R[2] * mz;
    const double b = tt[1] + R[3] * mx + R[4] * my + R[5] * mz;
    const double c = tt[2] + R[6] * mx + R[7] * my + R[8] * mz;
    tt[0] = scale * a;
    tt[1] = scale * b;
    tt[2] = scale * c;
  }
}
// X' = scale X - scale mean
__global__ void __launch_bounds__(kBlock)
    k_transform_pts(long P, double scale, double mx, double my, double mz, double* __restrict__ X) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    X[3 * p] = scale * X[3 * p] - scale * mx;
    X[3 * p + 1] = scale * X[3 * p + 1] - scale * my;
    X[3 * p + 2] = scale * X[3 * p + 2] - scale * mz;
  }
}

// FilterRotations: angle between R_j R_i^T and the measured R_ij (Eigen angularDistance), in degrees
__global__ void __launch_bounds__(kBlock)
    k_filter_rot(long E, const int* __restrict__ ei, const int* __restrict__ ej, const double* __restrict__ eq,
                 const double* __restrict__ nq, double max_deg, unsigned char* __restrict__ keep,
                 unsigned long long* __restrict__ counter) {
  unsigned long long c = 0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
    const Quat qi = load_quat(nq + 4 * (long)ei[e]), qj = load_quat(nq + 4 * (long)ej[e]);
    const Quat calc = qmul(qj, qconj(qi));
    const Quat d = qmul(qconj(calc), load_quat(eq + 4 * e));
    const double vn = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    const double ang = 2.0 * atan2(vn, fabs(d.w)) * (180.0 / M_PI);
    const bool bad = ang > max_deg;
    keep[e] = bad ? 0 : 1;
    c += bad ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(counter, c);
}

struct FilterWs {
  DevBuf<long> off;
  DevBuf<int> cam, cam_intr, intr_model, ei, ej, obs_pt;
  DevBuf<double> undist, xy, q, t, X, intr, eq, nq;
  DevBuf<float> cen;
  DevBuf<unsigned char> cal, keep, reg;
  DevBuf<unsigned long long> counter;
  DevBuf<long> cflag, cpos;             // gsfm_tracks_compact: survivor flags and their exclusive scan
  DevBuf<unsigned char> ctmp, cscan;    // ... scatter target, rocPRIM scratch
  static void destroy(void* p) { delete static_cast<FilterWs*>(p); }
};

FilterWs* filter_ws(gsfm_ctx* ctx) {
  if (!ctx->fl_ws) {
    ctx->fl_ws = new FilterWs();
    ctx->fl_ws_free = &FilterWs::destroy;
  }
  return static_cast<FilterWs*>(ctx->fl_ws);
}

// Arrays already resident in HBM are used in place; host arrays are staged into the ctx workspace.
template <typename T>
const T* dev_in(gsfm_ctx* ctx, DevBuf<T>& buf, const T* src, size_t n, int mem) {
  if (mem == GSFM_MEM_DEVICE) return src;
  T* dst = buf.ensure(n + 1);
  copy_in(ctx, dst, src, n, mem);
  return dst;
}

// Index sanity of the flat arrays (a malformed view must come back as GSFM_ERR_INVALID_ARGUMENT, not as out-of-bounds
// device reads): pt_offset starts at 0, is monotone and ends at M; every obs_cam / cam_intr / edge endpoint is in range.
__global__ void __launch_bounds__(kBlock)
    k_validate_view(long P, long M, int N, const long* __restrict__ off, const int* __restrict__ cam,
                    const int* __restrict__ cam_intr, int K, unsigned long long* __restrict__ bad) {
  const long stride = (long)gridDim.x * blockDim.x;
  bool err = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < M || i <= P || i < N; i += stride) {
    if (cam != nullptr && i < M && (cam[i] < 0 || cam[i] >= N)) err = true;
    if (i < P && (off[i] > off[i + 1] || off[i] < 0)) err = true;
    if (i == P && (off[P] != M || off[0] != 0)) err = true;
    if (cam_intr != nullptr && i < N && (cam_intr[i] < 0 || cam_intr[i] >= K)) err = true;
  }
  if (err) *bad = 1ull;
}
__global__ void __launch_bounds__(kBlock)
    k_validate_edges(long E, int N, const int* __restrict__ ei, const int* __restrict__ ej, unsigned long long* __restrict__ bad) {
  const long stride = (long)gridDim.x * blockDim.x;
  bool err = false;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < E; e += stride)
    if (ei[e] < 0 || ei[e] >= N || ej[e] < 0 || ej[e] >= N) err = true;
  if (err) *bad = 1ull;
}

// every idx[i] in [0, K)
__global__ void __launch_bounds__(kBlock)
    k_validate_index(long n, const int* __restrict__ idx, int K, unsigned long long* __restrict__ bad) {
  unsigned long long c = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    c += (idx[i] < 0 || idx[i] >= K) ? 1 : 0;
  if (c) atomicAdd(bad, c);
}

long read_counter(gsfm_ctx* ctx, FilterWs* ws);

void require_valid(gsfm_ctx* ctx, FilterWs* ws, const char* what) {
  GSFM_REQUIRE(read_counter(ctx, ws) == 0, what);
}

void stage_view(gsfm_ctx* ctx, FilterWs* ws, const gsfm_scene_view* v, bool need_undist, bool need_pixels, ViewDev& d) {
  GSFM_REQUIRE(v && v->pt_offset && v->obs_cam && v->cam_q && v->cam_t && v->pt_xyz, "filter: null argument");
  GSFM_REQUIRE(v->num_cams > 0 && v->num_pts >= 0 && v->num_obs >= 0, "filter: bad sizes");
  const int mem = v->mem;
  const long P = v->num_pts, M = v->num_obs;
  const int N = v->num_cams;
  d = ViewDev{};
  d.N = N;
  d.P = P;
  d.M = M;
  d.off = dev_in(ctx, ws->off, reinterpret_cast<const long*>(v->pt_offset), (size_t)P + 1, mem);
  d.cam = dev_in(ctx, ws->cam, v->obs_cam, (size_t)M, mem);
  d.q = dev_in(ctx, ws->q, v->cam_q, 4 * (size_t)N, mem);
  d.t = dev_in(ctx, ws->t, v->cam_t, 3 * (size_t)N, mem);
  d.X = dev_in(ctx, ws->X, v->pt_xyz, 3 * (size_t)P, mem);
  if (need_undist) {
    GSFM_REQUIRE(v->obs_undist != nullptr, "filter: obs_undist required");
    d.undist = dev_in(ctx, ws->undist, v->obs_undist, 3 * (size_t)M, mem);
  }
  if (need_pixels) {
    GSFM_REQUIRE(v->obs_xy && v->cam_intr && v->intr_model && v->intr_params && v->num_intr > 0,
                 "filter: pixel-space reprojection needs obs_xy and the intrinsics arrays");
    d.xy = dev_in(ctx, ws->xy, v->obs_xy, 2 * (size_t)M, mem);
    d.cam_intr = dev_in(ctx, ws->cam_intr, v->cam_intr, (size_t)N, mem);
    d.intr_model = dev_in(ctx, ws->intr_model, v->intr_model, (size_t)v->num_intr, mem);
    const int stride = v->intr_stride == 0 ? GSFM_CAMERA_MAX_PARAMS : v->intr_stride;
    GSFM_REQUIRE(stride == GSFM_CAMERA_MAX_PARAMS || stride == GSFM_CAMERA_MAX_PARAMS_WIDE, "filter: intr_stride must be 0, 8 or 16");
    d.intr_params = dev_in(ctx, ws->intr, v->intr_params, (size_t)stride * (size_t)v->num_intr, mem);
  }
  if (v->cam_calibrated) d.calibrated = dev_in(ctx, ws->cal, v->cam_calibrated, (size_t)N, mem);
  ws->counter.ensure(1);
  GSFM_HIP_CHECK(hipMemsetAsync(ws->counter.get(), 0, sizeof(unsigned long long), ctx->stream));
  const long span = std::max<long>(std::max<long>(M, P + 1), N);
  hipLaunchKernelGGL(k_validate_view, dim3(grid_wide(span, kBlock, 1 << 16)), dim3(kBlock), 0, ctx->stream, P, M, N, d.off,
                     d.cam, need_pixels ? d.cam_intr : nullptr, need_pixels ? v->num_intr : 0, ws->counter.get());
  require_valid(ctx, ws, "filter: pt_offset / obs_cam / cam_intr out of range");
}

long read_counter(gsfm_ctx* ctx, FilterWs* ws) {
  unsigned long long* h = reinterpret_cast<unsigned long long*>(ctx->h_pinned + 600);
  GSFM_HIP_CHECK(hipMemcpyAsync(h, ws->counter.get(), sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  GSFM_HIP_CHECK(hipGetLastError());
  return (long)h[0];
}

int filter_obs_impl(gsfm_ctx* ctx, const gsfm_scene_view* view, int mode, double thr, double thr2, uint8_t* keep_out,
                    int64_t* changed) {
  GSFM_REQUIRE(keep_out != nullptr, "filter: null output");
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  FilterWs* ws = filter_ws(ctx);
  ViewDev d;
  stage_view(ctx, ws, view, mode != 1, mode == 1, d);
  hipStream_t s = ctx->stream;
  const bool dev = view->mem == GSFM_MEM_DEVICE;
  unsigned char* keep = dev ? keep_out : ws->keep.ensure(d.M + 1);
  ws->counter.ensure(1);
  GSFM_HIP_CHECK(hipMemsetAsync(ws->counter.get(), 0, sizeof(unsigned long long), s));
  if (d.M > 0) {
    hipLaunchKernelGGL(k_fill_obs_pt, dim3(grid_wide(d.P, kBlock, 1 << 16)), dim3(kBlock), 0, s, d.P, d.off, ws->obs_pt.ensure(d.M + 1));
    bool wide = false;
    if (mode == 1) {  // which projection instance: a handful of model ids, read back with the validation flag's sync behind us
      std::vector<int> h_model;
      to_host(ctx, h_model, view->intr_model, (size_t)view->num_intr, view->mem);
      GSFM_HIP_CHECK(hipStreamSynchronize(s));
      for (int m : h_model) {
        wide = wide || m >= GSFM_CAMERA_OPENCV_FISHEYE;
        const bool known = m >= GSFM_CAMERA_SIMPLE_PINHOLE && m <= GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE;
        const bool needs16 = m == GSFM_CAMERA_FULL_OPENCV || m == GSFM_CAMERA_THIN_PRISM_FISHEYE || m == GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE;
        if (!known || (needs16 && view->intr_stride != GSFM_CAMERA_MAX_PARAMS_WIDE))
          throw StatusError(GSFM_ERR_UNSUPPORTED, needs16 ? "filter: camera models with more than 8 parameters need intr_stride = 16"
                                                          : "filter: camera model not supported");
      }
    }
    const bool wide16 = mode == 1 && view->intr_stride == GSFM_CAMERA_MAX_PARAMS_WIDE;
    const bool timed = ctx->prof.begin(s, GSFM_KERNEL_FILTER_OBS);
    if (wide16)
      hipLaunchKernelGGL((k_filter_obs<true, 16>), dim3(grid_wide(d.M, kBlock, 1 << 16)), dim3(kBlock), 0, s, d, ws->obs_pt.get(), mode, thr, thr2, keep);
    else if (wide)
      hipLaunchKernelGGL((k_filter_obs<true, 8>), dim3(grid_wide(d.M, kBlock, 1 << 16)), dim3(kBlock), 0, s, d, ws->obs_pt.get(), mode, thr, thr2, keep);
    else
      hipLaunchKernelGGL((k_filter_obs<false, 8>), dim3(grid_wide(d.M, kBlock, 1 << 16)), dim3(kBlock), 0, s, d, ws->obs_pt.get(), mode, thr, thr2, keep);
    if (timed) ctx->prof.end(s);
    hipLaunchKernelGGL(k_count_changed, dim3(grid_for(d.P, kBlock)), dim3(kBlock), 0, s, d.P, d.off, keep, ws->counter.get());
  }
  if (!dev) copy_out(ctx, keep_out, keep, (size_t)d.M, view->mem);
  const long c = read_counter(ctx, ws);
  if (changed) *changed = c;
  return GSFM_OK;
}


// ---- compaction after a filter (gsfm_tracks_compact) ---------------------------------------------------------------
// flag[k] = 1 when observation k survives its own keep flag and its track's; flag[M] = 0 (the scan's total slot)
__global__ void __launch_bounds__(kBlock)
    k_compact_flags(long M, const int* __restrict__ obs_pt, const unsigned char* __restrict__ obs_keep,
                    const unsigned char* __restrict__ track_keep, long* __restrict__ flag) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k <= M; k += (long)gridDim.x * blockDim.x) {
    long f = 0;
    if (k < M) f = ((obs_keep == nullptr || obs_keep[k] != 0) && (track_keep == nullptr || track_keep[obs_pt[k]] != 0)) ? 1 : 0;
    flag[k] = f;
  }
}
// new start of track p = number of surviving observations in front of its old start (pos[M] = total for p = P)
__global__ void __launch_bounds__(kBlock) k_compact_offsets(long P, const long* __restrict__ pos, long* __restrict__ off) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p <= P; p += (long)gridDim.x * blockDim.x) off[p] = pos[off[p]];
}
// one thread per (observation, 4-byte word): dst[pos[k]] = src[k] for the survivors
__global__ void __launch_bounds__(kBlock)
    k_compact_scatter(long M, int words, const long* __restrict__ flag, const long* __restrict__ pos,
                      const unsigned* __restrict__ src, unsigned* __restrict__ dst) {
  const long n = M * words;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long k = i / words;
    if (flag[k]) dst[pos[k] * words + (i - k * words)] = src[i];
  }
}
}  // namespace
}  // namespace gsfm

using namespace gsfm;

extern "C" int gsfm_filter_tracks_by_reprojection(gsfm_ctx* ctx, const gsfm_scene_view* view, double max_reprojection_error,
                                                  int in_normalized_image, uint8_t* obs_keep_out, int64_t* tracks_changed) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    return filter_obs_impl(ctx, view, in_normalized_image ? 0 : 1, max_reprojection_error, 0.0, obs_keep_out, tracks_changed);
  });
}

extern "C" int gsfm_filter_tracks_by_angle(gsfm_ctx* ctx, const gsfm_scene_view* view, double max_angle_error_deg,
                                           uint8_t* obs_keep_out, int64_t* tracks_changed) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    const double thr = std::cos(max_angle_error_deg * M_PI / 180.0);
    const double thr_u = std::cos(2.0 * max_angle_error_deg * M_PI / 180.0);  // track_filter.cc:61-62
    return filter_obs_impl(ctx, view, 2, thr, thr_u, obs_keep_out, tracks_changed);
  });
}

// ---- UndistortImages (processors/image_undistorter.cc:7-46) -----------------------------------------------------------
// features_undist[i] = camera.CamFromImg(features[i]).value_or(Zero).homogeneous().normalized(): the unit ray of a pixel.
// COLMAP's CamFromImg of a model with distortion is a Newton iteration on its own ImgFromCam (BaseCameraModel::
// IterativeUndistortion: at most 100 steps, stop at |step|^2 < 1e-10; COLMAP is un-vendored, its scheme restated).  Here:
// Newton on camera.hpp's distort_project — the projection the bundle adjustment uses, with its ANALYTIC Jacobian
// d pixel / d (u, v) where COLMAP differentiates numerically — from the pinhole guess J(0,0)^-1 (pixel - c), the same step
// limit and stopping rule.  Both iterations converge quadratically to the same root; what differs is below the stop's 1e-10.
// One lane per feature; rows of KP doubles (8, or 16 for the models with more than eight parameters).
template <int KP>
__global__ void __launch_bounds__(kBlock)
    k_undistort(long F, const double* __restrict__ xy, const int* __restrict__ feat_intr, const int* __restrict__ intr_model,
                const double* __restrict__ intr_params, double* __restrict__ rays) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < F; i += (long)gridDim.x * blockDim.x) {
    const int ik = feat_intr[i];
    const int model = intr_model[ik];
    double par[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) par[j] = j < KP ? intr_params[KP * (long)ik + j] : 0.0;
    const double ox = xy[2 * i], oy = xy[2 * i + 1];
    double px, py, J[4], Jp[2][16];
    distort_project<true, 16>(model, par, 0.0, 0.0, px, py, J, Jp);
    double det = J[0] * J[3] - J[1] * J[2];
    double u = (J[3] * (ox - px) - J[1] * (oy - py)) / det;
    double v = (J[0] * (oy - py) - J[2] * (ox - px)) / det;
    for (int it = 0; it < 100; ++it) {
      distort_project<true, 16>(model, par, u, v, px, py, J, Jp);
      det = J[0] * J[3] - J[1] * J[2];
      const double rx = px - ox, ry = py - oy;
      const double su = (J[3] * rx - J[1] * ry) / det, sv = (J[0] * ry - J[2] * rx) / det;
      u -= su;
      v -= sv;
      // (in the units of the normalised plane, as COLMAP's: its residual is x + dx(x) - x0)
      if (!(su * su + sv * sv >= 1e-10)) break;
    }
    V3 r{0.0, 0.0, 1.0};  // value_or(Zero).homogeneous().normalized()
    if (isfinite(u) && isfinite(v)) {
      const double inv = 1.0 / sqrt(u * u + v * v + 1.0);
      r = V3{u * inv, v * inv, inv};
    }
    st3(rays + 3 * i, r);
  }
}

extern "C" int gsfm_undistort_features(gsfm_ctx* ctx, int32_t mem, int64_t num_feat, const double* feat_xy, const int32_t* feat_intr,
                                       int32_t num_intr, const int32_t* intr_model, const double* intr_params, int32_t intr_stride,
                                       double* rays_out) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(num_feat >= 0 && num_intr > 0, "undistort: bad sizes");
    GSFM_REQUIRE(num_feat == 0 || (feat_xy && feat_intr && rays_out), "undistort: null argument");
    GSFM_REQUIRE(intr_model && intr_params, "undistort: null intrinsics");
    const int stride = intr_stride == 0 ? GSFM_CAMERA_MAX_PARAMS : intr_stride;
    GSFM_REQUIRE(stride == GSFM_CAMERA_MAX_PARAMS || stride == GSFM_CAMERA_MAX_PARAMS_WIDE, "undistort: intr_stride must be 0, 8 or 16");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    FilterWs* ws = filter_ws(ctx);
    hipStream_t s = ctx->stream;
    const long F = num_feat;
    if (F == 0) return (int)GSFM_OK;
    const double* d_xy = dev_in(ctx, ws->xy, feat_xy, 2 * (size_t)F, mem);
    const int* d_fi = dev_in(ctx, ws->cam, feat_intr, (size_t)F, mem);
    const int* d_model = dev_in(ctx, ws->intr_model, intr_model, (size_t)num_intr, mem);
    const double* d_par = dev_in(ctx, ws->intr, intr_params, (size_t)stride * (size_t)num_intr, mem);
    // index and model sanity before any gather (a malformed call comes back as INVALID_ARGUMENT / UNSUPPORTED)
    std::vector<int> h_model;
    to_host(ctx, h_model, intr_model, (size_t)num_intr, mem);
    for (int k = 0; k < num_intr; ++k) {
      GSFM_REQUIRE(h_model[k] >= 0 && h_model[k] <= GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE, "undistort: unknown camera model");
      const bool wide = h_model[k] == GSFM_CAMERA_FULL_OPENCV || h_model[k] == GSFM_CAMERA_THIN_PRISM_FISHEYE ||
                        h_model[k] == GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE;
      if (wide && stride != GSFM_CAMERA_MAX_PARAMS_WIDE)
        throw StatusError(GSFM_ERR_UNSUPPORTED, "undistort: camera models with more than 8 parameters need intr_stride = 16");
    }
    ws->counter.ensure(1);
    GSFM_HIP_CHECK(hipMemsetAsync(ws->counter.get(), 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_validate_index, dim3(grid_wide(F, kBlock, 1 << 16)), dim3(kBlock), 0, s, F, d_fi, num_intr, ws->counter.get());
    require_valid(ctx, ws, "undistort: feat_intr out of range");
    double* d_out = mem == GSFM_MEM_DEVICE ? rays_out : ws->undist.ensure(3 * (size_t)F + 3);
    const int grid = grid_wide(F, kBlock, 1 << 16);
    if (stride == GSFM_CAMERA_MAX_PARAMS)
      hipLaunchKernelGGL((k_undistort<8>), dim3(grid), dim3(kBlock), 0, s, F, d_xy, d_fi, d_model, d_par, d_out);
    else
      hipLaunchKernelGGL((k_undistort<16>), dim3(grid), dim3(kBlock), 0, s, F, d_xy, d_fi, d_model, d_par, d_out);
    if (mem != GSFM_MEM_DEVICE) copy_out(ctx, rays_out, d_out, 3 * (size_t)F, mem);
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    GSFM_HIP_CHECK(hipGetLastError());
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_filter_tracks_triangulation_angle(gsfm_ctx* ctx, const gsfm_scene_view* view, double min_angle_deg,
                                                      uint8_t* track_keep_out, int64_t* tracks_removed) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(track_keep_out != nullptr, "filter: null output");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    FilterWs* ws = filter_ws(ctx);
    ViewDev d;
    stage_view(ctx, ws, view, false, false, d);
    hipStream_t s = ctx->stream;
    ws->keep.ensure(d.P + 1);
    ws->counter.ensure(1);
    GSFM_HIP_CHECK(hipMemsetAsync(ws->counter.get(), 0, sizeof(unsigned long long), s));
    if (d.P > 0)
      hipLaunchKernelGGL(k_filter_tri, dim3(grid_wide(d.P, kBlock, 1 << 16)), dim3(kBlock), 0, s, d,
                         std::cos(min_angle_deg * M_PI / 180.0), ws->keep.get(), ws->counter.get());
    copy_out(ctx, track_keep_out, ws->keep.get(), (size_t)d.P, view->mem);
    const long c = read_counter(ctx, ws);
    if (tracks_removed) *tracks_removed = c;
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_normalize_reconstruction(gsfm_ctx* ctx, int32_t mem, int32_t num_cams, const uint8_t* cam_registered,
                                             const double* cam_q, double* cam_t_inout, int64_t num_pts, double* pt_xyz_inout,
                                             int32_t fixed_scale, double extent, double p0, double p1, double sim3_out[4]) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(cam_q && cam_t_inout && (num_pts == 0 || pt_xyz_inout) && num_cams > 0, "normalize: null argument");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    FilterWs* ws = filter_ws(ctx);
    hipStream_t s = ctx->stream;
    const int N = num_cams;
    const long P = num_pts;
    copy_in(ctx, ws->q.ensure(4 * (size_t)N), cam_q, 4 * (size_t)N, mem);
    copy_in(ctx, ws->t.ensure(3 * (size_t)N), cam_t_inout, 3 * (size_t)N, mem);
    copy_in(ctx, ws->X.ensure(3 * (size_t)P + 3), pt_xyz_inout, 3 * (size_t)P, mem);
    hipLaunchKernelGGL(k_centers_f32, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, s, N, ws->q.get(), ws->t.get(), ws->cen.ensure(3 * (size_t)N));
    // the percentile selection works on N floats per axis: host (N log N on a few thousand values)
    std::vector<float> cen(3 * (size_t)N);
    std::vector<unsigned char> reg;
    if (cam_registered) to_host(ctx, reg, cam_registered, (size_t)N, mem);
    GSFM_HIP_CHECK(hipMemcpyAsync(cen.data(), ws->cen.get(), cen.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<float> cx, cy, cz;
    for (int n = 0; n < N; ++n) {
      if (cam_registered && !reg[n]) continue;
      cx.push_back(cen[3 * (size_t)n]);
      cy.push_back(cen[3 * (size_t)n + 1]);
      cz.push_back(cen[3 * (size_t)n + 2]);
    }
    GSFM_REQUIRE(!cx.empty(), "normalize: no registered image");
    std::sort(cx.begin(), cx.end());
    std::sort(cy.begin(), cy.end());
    std::sort(cz.begin(), cz.end());
    const size_t n = cx.size();
    const size_t P0 = static_cast<size_t>((n > 3) ? p0 * (n - 1) : 0);
    const size_t P1 = static_cast<size_t>((n > 3) ? p1 * (n - 1) : n - 1);
    const double bmin[3] = {cx[P0], cy[P0], cz[P0]}, bmax[3] = {cx[P1], cy[P1], cz[P1]};
    double mean[3] = {0, 0, 0};
    for (size_t i = P0; i <= P1; ++i) {
      mean[0] += cx[i];
      mean[1] += cy[i];
      mean[2] += cz[i];
    }
    for (double& m : mean) m /= (double)(P1 - P0 + 1);
    double scale = 1.0;
    if (!fixed_scale) {
      const double old_extent = std::sqrt((bmax[0] - bmin[0]) * (bmax[0] - bmin[0]) + (bmax[1] - bmin[1]) * (bmax[1] - bmin[1]) +
                                          (bmax[2] - bmin[2]) * (bmax[2] - bmin[2]));
      if (old_extent >= std::numeric_limits<double>::epsilon()) scale = extent / old_extent;
    }
    hipLaunchKernelGGL(k_transform_cams, dim3(grid_for(N, kBlock)), dim3(kBlock), 0, s, N, scale, mean[0], mean[1], mean[2],
                       ws->q.get(), ws->t.get());
    if (P > 0)
      hipLaunchKernelGGL(k_transform_pts, dim3(grid_wide(P, kBlock, 1 << 16)), dim3(kBlock), 0, s, P, scale, mean[0], mean[1],
                         mean[2], ws->X.get());
    copy_out(ctx, cam_t_inout, ws->t.get(), 3 * (size_t)N, mem);
    copy_out(ctx, pt_xyz_inout, ws->X.get(), 3 * (size_t)P, mem);
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    GSFM_HIP_CHECK(hipGetLastError());
    if (sim3_out) {
      sim3_out[0] = scale;
      sim3_out[1] = -scale * mean[0];
      sim3_out[2] = -scale * mean[1];
      sim3_out[3] = -scale * mean[2];
    }
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_filter_rotations(gsfm_ctx* ctx, int32_t mem, int32_t num_nodes, const double* node_q, int64_t num_edges,
                                     const int32_t* edge_i, const int32_t* edge_j, const double* edge_q, double max_angle_deg,
                                     uint8_t* edge_keep_out, int64_t* num_invalid) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(node_q && edge_i && edge_j && edge_q && edge_keep_out && num_nodes > 0 && num_edges >= 0, "filter: null argument");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    FilterWs* ws = filter_ws(ctx);
    hipStream_t s = ctx->stream;
    const long E = num_edges;
    copy_in(ctx, ws->nq.ensure(4 * (size_t)num_nodes), node_q, 4 * (size_t)num_nodes, mem);
    copy_in(ctx, ws->ei.ensure(E + 1), edge_i, (size_t)E, mem);
    copy_in(ctx, ws->ej.ensure(E + 1), edge_j, (size_t)E, mem);
    copy_in(ctx, ws->eq.ensure(4 * (size_t)E + 4), edge_q, 4 * (size_t)E, mem);
    ws->keep.ensure(E + 1);
    ws->counter.ensure(1);
    GSFM_HIP_CHECK(hipMemsetAsync(ws->counter.get(), 0, sizeof(unsigned long long), s));
    if (E > 0) {
      hipLaunchKernelGGL(k_validate_edges, dim3(grid_wide(E, kBlock, 1 << 16)), dim3(kBlock), 0, s, E, num_nodes, ws->ei.get(),
                         ws->ej.get(), ws->counter.get());
      require_valid(ctx, ws, "filter: edge endpoint out of range");
    }
    if (E > 0)
      hipLaunchKernelGGL(k_filter_rot, dim3(grid_wide(E, kBlock, 1 << 16)), dim3(kBlock), 0, s, E, ws->ei.get(), ws->ej.get(),
                         ws->eq.get(), ws->nq.get(), max_angle_deg, ws->keep.get(), ws->counter.get());
    copy_out(ctx, edge_keep_out, ws->keep.get(), (size_t)E, mem);
    const long c = read_counter(ctx, ws);
    if (num_invalid) *num_invalid = c;
    return (int)GSFM_OK;
  });
}

// Compaction after a filter.  The reference erases the dropped observations from Track::observations (track_filter.cc:36-44,
// 75-83) or clears the list of a whole track (:120-123); in the flat layout the survivors move up, in order.  Device memory
// stays on the device — only the new observation count is read back — which is what lets the BA outer loop of
// global_mapper.cc:201-275 (solve, solve, normalise, filter, and again) run without the per-call pack / unpack.
extern "C" int gsfm_tracks_compact(gsfm_ctx* ctx, int32_t mem, int64_t num_pts, int64_t num_obs, int64_t* pt_offset_inout,
                                   const uint8_t* obs_keep, const uint8_t* track_keep, int32_t num_arrays,
                                   void* const* arrays_inout, const int32_t* elem_bytes, int64_t* num_obs_out) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    const long P = num_pts, M = num_obs;
    GSFM_REQUIRE(pt_offset_inout != nullptr && P >= 0 && M >= 0 && num_arrays >= 0 && num_arrays <= 16, "compact: bad argument");
    GSFM_REQUIRE(num_arrays == 0 || (arrays_inout != nullptr && elem_bytes != nullptr), "compact: null array table");
    for (int a = 0; a < num_arrays; ++a)
      GSFM_REQUIRE((arrays_inout[a] != nullptr || M == 0) && elem_bytes[a] > 0 && elem_bytes[a] % 4 == 0,
                   "compact: per-observation arrays of 4-byte words");
    long* off = reinterpret_cast<long*>(pt_offset_inout);
    if (mem != GSFM_MEM_DEVICE) {  // host arrays: a plain loop (nothing here is worth a round trip over PCIe)
      GSFM_REQUIRE(off[0] == 0 && off[P] == M, "compact: pt_offset does not span the observations");
      for (long p = 0; p < P; ++p)  // validate BEFORE the first write: a malformed table must leave the caller's arrays untouched
        GSFM_REQUIRE(off[p] <= off[p + 1] && off[p + 1] <= M, "compact: pt_offset not monotone");
      long w = 0;
      for (long p = 0; p < P; ++p) {
        const long k0 = off[p], k1 = off[p + 1];
        off[p] = w;
        for (long k = k0; k < k1; ++k) {
          if ((obs_keep != nullptr && obs_keep[k] == 0) || (track_keep != nullptr && track_keep[p] == 0)) continue;
          if (w != k)
            for (int a = 0; a < num_arrays; ++a)
              std::memmove(static_cast<char*>(arrays_inout[a]) + (size_t)w * elem_bytes[a],
                           static_cast<const char*>(arrays_inout[a]) + (size_t)k * elem_bytes[a], (size_t)elem_bytes[a]);
          ++w;
        }
      }
      off[P] = w;
      if (num_obs_out) *num_obs_out = w;
      return (int)GSFM_OK;
    }
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    FilterWs* ws = filter_ws(ctx);
    hipStream_t s = ctx->stream;
    // index sanity first: a malformed pt_offset must not become an out-of-bounds scatter
    ws->counter.ensure(1);
    GSFM_HIP_CHECK(hipMemsetAsync(ws->counter.get(), 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_validate_view, dim3(grid_wide(std::max<long>(M, P + 1), kBlock, 1 << 16)), dim3(kBlock), 0, s, P, M, 1,
                       (const long*)off, (const int*)nullptr, (const int*)nullptr, 0, ws->counter.get());
    require_valid(ctx, ws, "compact: pt_offset out of range");
    long total = 0;
    if (M > 0) {
      int* obs_pt = ws->obs_pt.ensure(M + 1);
      hipLaunchKernelGGL(k_fill_obs_pt, dim3(grid_wide(P, kBlock, 1 << 16)), dim3(kBlock), 0, s, P, (const long*)off, obs_pt);
      long* flag = ws->cflag.ensure(M + 2);
      long* pos = ws->cpos.ensure(M + 2);
      hipLaunchKernelGGL(k_compact_flags, dim3(grid_wide(M + 1, kBlock, 1 << 16)), dim3(kBlock), 0, s, M, (const int*)obs_pt, obs_keep,
                         track_keep, flag);
      exclusive_scan_i64(ctx, ws->cscan, flag, pos, (size_t)M + 1);
      for (int a = 0; a < num_arrays; ++a) {
        const int words = elem_bytes[a] / 4;
        unsigned* tmp = reinterpret_cast<unsigned*>(ws->ctmp.ensure((size_t)M * elem_bytes[a] + 16));
        hipLaunchKernelGGL(k_compact_scatter, dim3(grid_wide(M * words, kBlock, 1 << 16)), dim3(kBlock), 0, s, M, words, (const long*)flag,
                           (const long*)pos, static_cast<const unsigned*>(arrays_inout[a]), tmp);
        // (the whole prefix is copied back: its length is known on the device only; the tail beyond the new count is dead)
        GSFM_HIP_CHECK(hipMemcpyAsync(arrays_inout[a], tmp, (size_t)M * elem_bytes[a], hipMemcpyDeviceToDevice, s));
      }
      hipLaunchKernelGGL(k_compact_offsets, dim3(grid_wide(P + 1, kBlock, 1 << 16)), dim3(kBlock), 0, s, P, (const long*)pos, off);
      long* h = reinterpret_cast<long*>(ctx->h_pinned + 616);
      GSFM_HIP_CHECK(hipMemcpyAsync(h, pos + M, sizeof(long), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipStreamSynchronize(s));
      GSFM_HIP_CHECK(hipGetLastError());
      total = h[0];
    }
    if (num_obs_out) *num_obs_out = total;
    return (int)GSFM_OK;
  });
}
