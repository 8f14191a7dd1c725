// obsgraph.hpp — the observation graph (tracks x cameras) in BOTH orders, shared by gp.hip / ba.hip.
//
// The reference hands Ceres one residual block per observation and lets SPARSE_SCHUR sort things
// out (global_positioning.cc:270-375, bundle_adjustment.cc:115-190).  Here the bipartite graph is
// laid out twice in HBM so that every reduction is a contiguous, atomic-free, fixed-order sum:
//
//   track-major  (the input order): obs of track p are [off[p], off[p+1]).  Point-side sums
//                (H_pp, g_p, the implicit-Schur "t_p = H_pp^-1 sum_k ..." of PCG phase A) run one
//                LANE PER OBSERVATION with a segmented wave scan; waves are cut at track
//                boundaries by the host-built tile table (<= 64 observations per tile, or one long
//                track per tile), so no segment ever straddles a wave.
//   camera-major (derived once per solve by a stable device radix sort on the camera index):
//                obs of camera n are [coff[n], coff[n+1]).  Camera-side sums (gradient, S_cc
//                blocks, PCG phase B "y_n = sum_k ...") run one WAVE PER CAMERA SEGMENT with the
//                camera data in registers and a single wave reduction at the end.  A segment is a
//                camera's whole list when it is short (the common case: nothing else happens) or
//                a slice of at most `seg_len` observations of a long one: real images differ by
//                10-100x in how many tracks they see, and one wave walking a 13 000-observation
//                list would set the duration of the whole sweep.  The slices of a long camera put
//                their partial sums into fixed slots and a second, tiny launch of the same kernel
//                (pass 1, one wave per cut camera, only when there are cut cameras) adds them in
//                slice order and finishes the camera — no floating-point atomics, a fixed
//                summation order, and no device-scope fences inside the sweep (a ticket / last-
//                arriver variant was measured first: its fences invalidate the XCD's L2 and made
//                the skewed sweep 3x slower than the uniform one).
//
// Observations of tracks shorter than min_num_view_per_track (gp.cc:258, ba.cc:122) are excluded
// from the camera-major lists (key = N sorts them behind every camera) and contribute zero in
// track-major sweeps.
#pragma once

#include <cstdlib>
#include <vector>

#include "device.hpp"

namespace gsfm {

void sort_pairs_i32(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const int* keys_in, int* keys_out,
                    const int* vals_in, int* vals_out, size_t n, int key_bits);

struct ObsGraph {  // device view, passed to kernels by value
  int N = 0;       // cameras
  int T = 0;       // wave tiles
  long P = 0, M = 0;
  long Mu = 0;     // observations of used tracks (= coff[N])
  const long* off = nullptr;           // [P+1]
  const int* cam = nullptr;            // [M]
  const unsigned char* used = nullptr; // [P]
  const int* obs_pt = nullptr;         // [M]   track of each observation
  const int* tile = nullptr;           // [T+1] first track of each wave tile
  const int* tile_k = nullptr;         // [T+1] first observation of each wave tile
  const int* coff = nullptr;           // [N+2] camera-major CSR; coff[N] = Mu, coff[N+1] = M
  const int* c_src = nullptr;          // [M]   track-major index of each camera-major slot
  const int* c_pt = nullptr;           // [M]   track of each camera-major slot
  // camera segments (see the header comment): S >= N segments in camera order
  int S = 0;                           // number of segments
  int nmulti = 0;                      // cameras cut into more than one segment (<= kMaxMultiCams)
  const int* seg_cam = nullptr;        // [S]   camera of the segment
  const int* seg_k = nullptr;          // [S+1] first camera-major slot of the segment (seg_k[S] = Mu)
  const int* seg_first = nullptr;      // [S]   first segment of this segment's camera
  const int* seg_cnt = nullptr;        // [S]   number of segments of this segment's camera
  const int* seg_multi = nullptr;      // [S]   slot of the segment among the slices of cut cameras (consecutive per camera), -1 for whole cameras
  const int* multi_first = nullptr;    // [nmulti] first segment of each cut camera
  double* segpart = nullptr;           // [#slices of cut cameras][W] parked partial sums (slot: seg_multi; W = the kernel's accumulator width)
  int pass = 0;                        // 0: sweep over the segments; 1: combine pass over the cut cameras (see cam_seg_*)
};

constexpr int kSegPartW = 192;     // widest per-camera accumulator (k_ba_build_cam<false, ., 16> of the 16-wide BA unit: 179 values; 8-wide joint: 119)
constexpr int kSegLenMin = 1024;   // observations per segment: 16 trips of a wave
constexpr int kMaxMultiCams = 1024;

struct ObsGraphWs {
  DevBuf<int> obs_pt, tile, tile_k, keys, keys_sorted, vals, c_src, c_pt, coff, flag;
  DevBuf<int> seg_cam, seg_k, seg_first, seg_cnt, seg_multi, multi_first;
  DevBuf<double> segpart;
  DevBuf<unsigned char> used, sort_tmp;
  std::vector<int> h_coff;  // host copy of the camera-major offsets
};

// ---- device helpers --------------------------------------------------------------------------
// Inclusive segmented scan over the 64 lanes of a wave: lanes with equal `key` must be contiguous.
// After the call the LAST lane of every segment holds the segment total (fixed summation order).
template <int K>
__device__ __forceinline__ void seg_scan(double (&v)[K], int key, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int ku = __shfl_up(key, d, 64);
    const bool take = lane >= d && ku == key;
    // no segment reaches back d lanes => none reaches further: the remaining steps add nothing (tracks are short, most
    // tiles finish after 3 of the 6 steps; the sums and their order are unchanged)
    if (__ballot(take) == 0ull) break;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const double vu = __shfl_up(v[k], d, 64);
      if (take) v[k] += vu;
    }
  }
}
__device__ __forceinline__ bool seg_is_tail(int key, int lane) {
  const int kn = __shfl_down(key, 1, 64);
  return lane == 63 || kn != key;
}

// Wave-wide sum of K values, result in every lane (xor butterfly, fixed order).
template <int K>
__device__ __forceinline__ void wave_allsum(double (&v)[K]) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += __shfl_xor(v[k], off, 64);
  }
}

// Camera-major kernels run as   for (it = wave; it < cam_seg_count(g); it += nwaves) { sg = cam_seg_index(g, it); ... }
// Pass 0 walks all segments; a cut camera's slices only park their wave sums (cam_seg_total returns false).  Pass 1
// walks the cut cameras: the observation loop is empty there (cam_seg_k0 == cam_seg_k1) and cam_seg_total loads the
// parked sums in slice order into lane 0's `acc`, after which the kernel's own per-camera epilogue runs unchanged.
__device__ __forceinline__ int cam_seg_count(const ObsGraph& g) { return g.pass == 0 ? g.S : g.nmulti; }
__device__ __forceinline__ int cam_seg_index(const ObsGraph& g, int it) { return g.pass == 0 ? it : g.multi_first[it]; }
__device__ __forceinline__ int cam_seg_k0(const ObsGraph& g, int sg) { return g.seg_k[sg]; }
__device__ __forceinline__ int cam_seg_k1(const ObsGraph& g, int sg) { return g.pass == 0 ? g.seg_k[sg + 1] : g.seg_k[sg]; }

// Returns true when lane 0's `acc` holds the camera's complete sum (call after wave_allsum).
// Parked sums are W doubles per slot (the kernel's own accumulator width: pass 0 and pass 1 of a kernel agree on it).  In
// pass 1 lane c adds column c of the camera's slots in slice order (coalesced rows); the totals are handed to `acc` with
// readlane — the same sums in the same order as a serial loop, without one lane walking cnt x W values, and without
// touching the slots.  Two things here are about the register allocator, not the arithmetic (tests/test_kernel_resources.py):
// broadcasting the totals into every lane's `acc` cost k_ba_build_cam its second wave per SIMD (251 -> 404 registers), and
// plain stores in both passes get merged by the compiler into one store through a select of a global and a private
// pointer, which pins `acc` in scratch — hence the atomic (relaxed, same instruction) stores when parking.
template <int W>
__device__ __forceinline__ bool cam_seg_total(const ObsGraph& g, int sg, double (&acc)[W], int lane) {
  static_assert(W <= kSegPartW, "accumulator wider than the partial-sum slots");
  const int cnt = g.seg_cnt[sg];
  if (g.pass == 0) {
    if (cnt == 1) return true;
    if (lane == 0) {
      double* dst = g.segpart + (size_t)g.seg_multi[sg] * W;
#pragma unroll
      for (int j = 0; j < W; ++j) __hip_atomic_store(dst + j, acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return false;
  }
  // pass 1 (sg = the camera's first segment) only READS the parked slots: it can be repeated, and nothing depends on pass 0
  // and pass 1 sharing more than the slot width W.  Lane c sums column c (and c + 64) over the slices in slice order — the
  // same sums in the same order as a serial loop — and the totals reach `acc` through readlane (wave-uniform values: no
  // memory round trip, no fence; until round 5 the totals overwrote the first slot and lane 0 read the row back).
  const double* src = g.segpart + (size_t)g.seg_multi[sg] * W;
  constexpr int TR = (W + 63) / 64;
  double col[TR];
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    const int c = lane + 64 * t;
    col[t] = 0.0;
    if (c < W)
      for (int q = 0; q < cnt; ++q) col[t] += src[(size_t)q * W + c];
  }
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const double v = col[j / 64];
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j % 64);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j % 64);
    if (lane == 0) acc[j] = __hiloint2double(hi, lo);  // (every lane taking the totals costs k_ba_build_cam 60 registers)
  }
  return true;
}

// ---- kernels ------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(kBlock)
    k_og_keys(int N, long P, const long* __restrict__ off, const int* __restrict__ cam,
              const unsigned char* __restrict__ used, int* __restrict__ obs_pt, int* __restrict__ keys,
              int* __restrict__ vals, int* __restrict__ flag) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const bool u = used[p] != 0;
    for (long k = off[p]; k < off[p + 1]; ++k) {
      const int n = cam[k];
      if (n < 0 || n >= N) atomicOr(flag, 1);
      obs_pt[k] = (int)p;
      keys[k] = u ? n : N;
      vals[k] = (int)k;
    }
  }
}

// coff[c] = first sorted slot whose key >= c, for c in [0, N+1]
static __global__ void __launch_bounds__(kBlock)
    k_og_offsets(int N, long M, const int* __restrict__ keys_sorted, int* __restrict__ coff) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i <= M; i += (long)gridDim.x * blockDim.x) {
    const int a = i == 0 ? -1 : keys_sorted[i - 1];
    const int b = i == M ? N + 1 : keys_sorted[i];
    for (int c = a + 1; c <= b; ++c) coff[c] = (int)i;
  }
}

static __global__ void __launch_bounds__(kBlock)
    k_og_gather_i32(long n, const int* __restrict__ idx, const int* __restrict__ src, int* __restrict__ dst) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[idx[i]];
}

// dst[i][0..W) = src[idx[i]][0..W)   (permuted copy of a per-observation array, W doubles wide)
template <int W>
static __global__ void __launch_bounds__(kBlock)
    k_og_gather_f64(long n, const int* __restrict__ idx, const double* __restrict__ src, double* __restrict__ dst) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long s = idx[i];
#pragma unroll
    for (int j = 0; j < W; ++j) dst[W * i + j] = src[W * s + j];
  }
}
static __global__ void __launch_bounds__(kBlock)
    k_og_gather_u8(long n, const int* __restrict__ idx, const unsigned char* __restrict__ src,
                   unsigned char* __restrict__ dst) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[idx[i]];
}

// ---- host ---------------------------------------------------------------------------------------
// h_off: host copy of pt_offset.  d_off / d_cam: device copies.  Fills `g`, returns the number of
// used observations; *first_used_obs receives the first observation of the first used track (-1 if
// none) — the reference's "first scale constant" gauge (gp.cc:484-489).
inline long build_obs_graph(gsfm_ctx* ctx, ObsGraphWs& ws, int N, long P, long M, const std::vector<long>& h_off,
                            const long* d_off, const int* d_cam, int min_len, ObsGraph& g, long* first_used_obs) {
  GSFM_REQUIRE(M < (1L << 31) - 64, "observation count must fit int32");
  hipStream_t s = ctx->stream;
  std::vector<unsigned char> h_used(P);
  std::vector<int> h_tile;
  h_tile.reserve((size_t)(M / 48 + 16));
  long m_used = 0, first = -1;
  long cur = 0;  // observations in the open tile
  bool open = false;
  for (long p = 0; p < P; ++p) {
    const long len = h_off[p + 1] - h_off[p];
    GSFM_REQUIRE(len >= 0, "pt_offset must be non-decreasing");
    h_used[p] = len >= min_len ? 1 : 0;
    if (h_used[p]) {
      m_used += len;
      if (first < 0) first = h_off[p];
    }
    if (len > 64) {
      h_tile.push_back((int)p);  // a long track is a tile of its own
      open = false;
      cur = 0;
    } else if (!open || cur + len > 64) {
      h_tile.push_back((int)p);
      open = true;
      cur = len;
    } else {
      cur += len;
    }
  }
  const int T = (int)h_tile.size();
  h_tile.push_back((int)P);
  std::vector<int> h_tile_k(T + 1);
  for (int i = 0; i <= T; ++i) h_tile_k[i] = (int)h_off[h_tile[i]];
  if (first_used_obs) *first_used_obs = first;

  GSFM_HIP_CHECK(hipMemcpyAsync(ws.used.ensure(P + 1), h_used.data(), (size_t)P, hipMemcpyHostToDevice, s));
  GSFM_HIP_CHECK(hipMemcpyAsync(ws.tile.ensure(T + 2), h_tile.data(), (size_t)(T + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  GSFM_HIP_CHECK(hipMemcpyAsync(ws.tile_k.ensure(T + 2), h_tile_k.data(), (size_t)(T + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  ws.obs_pt.ensure(M + 1);
  ws.keys.ensure(M + 1);
  ws.keys_sorted.ensure(M + 1);
  ws.vals.ensure(M + 1);
  ws.c_src.ensure(M + 1);
  ws.c_pt.ensure(M + 1);
  ws.coff.ensure(N + 3);
  ws.flag.ensure(4);
  GSFM_HIP_CHECK(hipMemsetAsync(ws.flag.get(), 0, 4 * sizeof(int), s));
  hipLaunchKernelGGL(k_og_keys, dim3(grid_for(P, kBlock)), dim3(kBlock), 0, s, N, P, d_off, d_cam, ws.used.get(),
                     ws.obs_pt.get(), ws.keys.get(), ws.vals.get(), ws.flag.get());
  int bits = 1;
  while ((1L << bits) <= N) ++bits;
  sort_pairs_i32(ctx, ws.sort_tmp, ws.keys.get(), ws.keys_sorted.get(), ws.vals.get(), ws.c_src.get(), (size_t)M, bits);
  hipLaunchKernelGGL(k_og_offsets, dim3(grid_for(M + 1, kBlock)), dim3(kBlock), 0, s, N, M, ws.keys_sorted.get(),
                     ws.coff.get());
  hipLaunchKernelGGL(k_og_gather_i32, dim3(grid_for(M, kBlock)), dim3(kBlock), 0, s, M, ws.c_src.get(),
                     ws.obs_pt.get(), ws.c_pt.get());
  int* h_flag = reinterpret_cast<int*>(ctx->h_pinned + 512);
  GSFM_HIP_CHECK(hipMemcpyAsync(h_flag, ws.flag.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  ws.h_coff.resize((size_t)N + 2);
  GSFM_HIP_CHECK(hipMemcpyAsync(ws.h_coff.data(), ws.coff.get(), (size_t)(N + 2) * sizeof(int), hipMemcpyDeviceToHost, s));
  GSFM_HIP_CHECK(hipStreamSynchronize(s));  // host vectors go out of scope; flag is read
  GSFM_REQUIRE(h_flag[0] == 0, "obs_cam out of range");
  // camera segments: the shortest power-of-two slice length >= kSegLenMin that cuts at most kMaxMultiCams cameras
  {
    const std::vector<int>& co = ws.h_coff;
    int seg_len = kSegLenMin;
    if (ctx->knob[GSFM_KNOB_SEG_LEN] > 0) seg_len = std::max(64, ctx->knob[GSFM_KNOB_SEG_LEN]);  // diagnostics / A-B runs
    for (;;) {
      int cut = 0;
      for (int n = 0; n < N; ++n) cut += (co[n + 1] - co[n] > seg_len) ? 1 : 0;
      if (cut <= kMaxMultiCams) break;
      seg_len *= 2;
    }
    // host layout: 5 arrays back to back (cam | k | first | cnt | multi), S + 1 entries each
    std::vector<int> cam, k0, first, cnt, multi, mfirst;
    int nmulti = 0, nslots = 0;
    for (int n = 0; n < N; ++n) {
      const int len = co[n + 1] - co[n];
      const int c = len > seg_len ? (len + seg_len - 1) / seg_len : 1;
      const int f = (int)cam.size();
      for (int q = 0; q < c; ++q) {
        cam.push_back(n);
        k0.push_back(co[n] + (int)(((long)len * q) / c));  // equal slices
        first.push_back(f);
        cnt.push_back(c);
        multi.push_back(c > 1 ? nslots++ : -1);
      }
      if (c > 1) {
        mfirst.push_back(f);
        ++nmulti;
      }
    }
    const int S = (int)cam.size();
    k0.push_back(co[N]);
    GSFM_HIP_CHECK(hipMemcpyAsync(ws.seg_cam.ensure(S + 1), cam.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws.seg_k.ensure(S + 2), k0.data(), (size_t)(S + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws.seg_first.ensure(S + 1), first.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws.seg_cnt.ensure(S + 1), cnt.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws.seg_multi.ensure(S + 1), multi.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice, s));
    ws.multi_first.ensure(nmulti + 1);
    if (nmulti > 0)
      GSFM_HIP_CHECK(hipMemcpyAsync(ws.multi_first.get(), mfirst.data(), (size_t)nmulti * sizeof(int), hipMemcpyHostToDevice, s));
    ws.segpart.ensure(std::max((size_t)1, (size_t)nslots * kSegPartW));  // only slices of cut cameras park sums
    GSFM_HIP_CHECK(hipStreamSynchronize(s));  // the host vectors above go out of scope
    g.S = S;
    g.nmulti = nmulti;
    g.seg_cam = ws.seg_cam.get();
    g.seg_k = ws.seg_k.get();
    g.seg_first = ws.seg_first.get();
    g.seg_cnt = ws.seg_cnt.get();
    g.seg_multi = ws.seg_multi.get();
    g.multi_first = ws.multi_first.get();
    g.segpart = ws.segpart.get();
    g.pass = 0;
  }
  g.N = N;
  g.T = T;
  g.P = P;
  g.M = M;
  g.Mu = m_used;
  g.off = d_off;
  g.cam = d_cam;
  g.used = ws.used.get();
  g.obs_pt = ws.obs_pt.get();
  g.tile = ws.tile.get();
  g.tile_k = ws.tile_k.get();
  g.coff = ws.coff.get();
  g.c_src = ws.c_src.get();
  g.c_pt = ws.c_pt.get();
  return m_used;
}

// ---- the chunked order of the camera-side sweeps --------------------------------------------------------------------
// A camera-major sweep gathers one point record per observation, and on scenes without locality every gather is an L2 miss:
// the sweep runs at the ~54 G line requests/s of the fabric whatever the record size (tools/exp_gather_calib.hip; 6.0 M
// gathers = 111 us, k_gp_phaseB measured 118).  The third copy of the observation graph turns those misses into hits
// (tools/exp_chunk_gather.hip: 116 -> 65 us for the same gathers): the tracks are cut into C = 8 R index ranges ("chunks")
// of at most ~2.7 MB of point records, the used camera-major slots are re-sorted by (chunk, camera) — stable, so points
// stay ascending inside a (chunk, camera) run — and laid out in 64-slot tiles, every chunk starting on a tile boundary.
// The tile list is cut into 8 equal parts and workgroup b works on part b % 8: with the observed dispatch rule (block b on
// XCD b % 8; used for speed only, any placement gives the same sums) every XCD walks its own R chunks one after the other
// and the records it gathers are resident in its own 4 MB L2.
// One lane per slot; per-camera sums are a wave segmented scan over the camera key.  A (camera, tile) run is a "piece":
// its last lane writes the piece's partial sum to out[slot] = position of the piece in (camera, slot) order, so that the
// per-camera totals are one contiguous, fixed-order sum over piece_off[n] .. piece_off[n + 1] (k_*_wsum kernels).
struct ObsX {  // device view
  int tiles = 0;                      // 64-slot tiles
  int per = 0;                        // tiles per part: part x = tiles [x per, (x + 1) per)
  int npieces = 0;
  const int2* ix = nullptr;           // [tiles * 64] (track, camera); (-1, 0): padding
  const int* src = nullptr;           // [tiles * 64] camera-major slot of the entry (-1: padding)
  const int* out = nullptr;           // [tiles * 64] piece index, valid in the last lane of a piece
  const int* piece_off = nullptr;     // [N + 2] pieces of camera n: piece_off[n] .. piece_off[n + 1]
  const int* c_xslot = nullptr;       // [Mu] slot of every used camera-major entry
};
struct ObsXWs {
  DevBuf<int> key, key_sorted, val, src_sorted, choff, xbase, src, out, piece_off, c_xslot, tkey, tkey_sorted, tval, tval_sorted;
  DevBuf<int2> ix;
  int chunks = 0;
};

// wave tile of this workgroup's wave in the XCD-affine walk (-1: none)
__device__ __forceinline__ int x_tile_of_wave(const ObsX& x) {
  const int j = (int)(blockIdx.x >> 3) * (kBlock / 64) + (int)(threadIdx.x >> 6);
  if (j >= x.per) return -1;
  const int t = (int)(blockIdx.x & 7) * x.per + j;
  return t < x.tiles ? t : -1;
}
inline int x_grid(const ObsX& x, int tiles_per_wave = 1) {
  const int waves = (x.per + tiles_per_wave - 1) / tiles_per_wave;  // per part
  return 8 * ((waves + kBlock / 64 - 1) / (kBlock / 64));
}
// last lane of a piece (key: camera of a valid slot, distinct negative values on padding)
__device__ __forceinline__ bool x_piece_tail(int key, int lane) { return seg_is_tail(key, lane) && key >= 0; }

static __global__ void __launch_bounds__(kBlock)
    k_ox_keys(long n, long pc, const int* __restrict__ c_pt, int* __restrict__ key, int* __restrict__ val) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    key[i] = (int)(c_pt[i] / pc);
    val[i] = (int)i;
  }
}
static __global__ void __launch_bounds__(kBlock)
    k_ox_fill(long n, const int* __restrict__ key_sorted, const int* __restrict__ src_sorted, const int* __restrict__ choff,
              const int* __restrict__ xbase, const int* __restrict__ c_pt, const int* __restrict__ c_cam,
              int2* __restrict__ ix, int* __restrict__ src, int* __restrict__ c_xslot) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = key_sorted[i], k = src_sorted[i];
    const int slot = xbase[c] + (int)(i - choff[c]);
    ix[slot] = make_int2(c_pt[k], c_cam[k]);
    src[slot] = k;
    c_xslot[k] = slot;
  }
}
static __global__ void __launch_bounds__(kBlock)
    k_ox_tails(long nslots, int N, const int2* __restrict__ ix, int* __restrict__ tkey, int* __restrict__ tval) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (long)gridDim.x * blockDim.x) {
    const int2 a = ix[i];
    bool tail = false;
    if (a.x >= 0) {
      if ((i & 63) == 63) {
        tail = true;
      } else {
        const int2 b = ix[i + 1];
        tail = b.x < 0 || b.y != a.y;
      }
    }
    tkey[i] = tail ? a.y : N;
    tval[i] = (int)i;
  }
}
static __global__ void __launch_bounds__(kBlock)
    k_ox_out(long npieces, const int* __restrict__ tval_sorted, int* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npieces; i += (long)gridDim.x * blockDim.x) out[tval_sorted[i]] = (int)i;
}

// Builds the chunked order from the camera-major one (build_obs_graph first).  rec_bytes: bytes of point record a sweep
// gathers per observation (sizes the chunks).  Returns false (and leaves x empty) when there is nothing to order.
inline bool build_x_order(gsfm_ctx* ctx, ObsGraphWs& ws, ObsXWs& xw, const ObsGraph& g, int rec_bytes, ObsX& x, int force_chunks = 0) {
  x = ObsX{};
  const long Mu = ws.h_coff[(size_t)g.N];
  if (Mu <= 0 || g.P <= 0) return false;
  hipStream_t s = ctx->stream;
  // chunks: C = 8 R, R such that a chunk's records are <= ~2.7 MB (measured optimum 2 - 4 MB at 64-byte records)
  const double per_xcd = (double)g.P * rec_bytes / 8.0;
  int R = (int)std::ceil(per_xcd / (2.75 * 1024 * 1024));
  R = std::max(1, std::min(R, 32));
  if (force_chunks >= 8) R = std::min(32, force_chunks / 8);  // A/B runs (GSFM_KNOB_CHUNKED_SWEEPS >= 8)
  const int C = 8 * R;
  xw.chunks = C;
  const long pc = (g.P + C - 1) / C;
  int bits = 1;
  while ((1 << bits) < C) ++bits;
  xw.key.ensure(Mu + 1);
  xw.key_sorted.ensure(Mu + 1);
  xw.val.ensure(Mu + 1);
  xw.src_sorted.ensure(Mu + 1);
  xw.choff.ensure(C + 3);
  xw.xbase.ensure(C + 3);
  xw.c_xslot.ensure(Mu + 1);
  hipLaunchKernelGGL(k_ox_keys, dim3(grid_for(Mu, kBlock)), dim3(kBlock), 0, s, Mu, pc, g.c_pt, xw.key.get(), xw.val.get());
  sort_pairs_i32(ctx, ws.sort_tmp, xw.key.get(), xw.key_sorted.get(), xw.val.get(), xw.src_sorted.get(), (size_t)Mu, bits);
  hipLaunchKernelGGL(k_og_offsets, dim3(grid_for(Mu + 1, kBlock)), dim3(kBlock), 0, s, C - 1, Mu, xw.key_sorted.get(), xw.choff.get());
  std::vector<int> h_choff((size_t)C + 1), h_base((size_t)C + 1);
  GSFM_HIP_CHECK(hipMemcpyAsync(h_choff.data(), xw.choff.get(), (size_t)(C + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
  GSFM_HIP_CHECK(hipStreamSynchronize(s));
  long nslots = 0;
  for (int c = 0; c < C; ++c) {
    h_base[c] = (int)nslots;
    nslots += ((long)(h_choff[c + 1] - h_choff[c]) + 63) / 64 * 64;  // every chunk starts on a tile boundary
  }
  h_base[C] = (int)nslots;
  GSFM_REQUIRE(nslots < (1L << 31) - 64, "observation count must fit int32");
  GSFM_HIP_CHECK(hipMemcpyAsync(xw.xbase.get(), h_base.data(), (size_t)(C + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  xw.ix.ensure(nslots + 64);
  xw.src.ensure(nslots + 64);
  xw.out.ensure(nslots + 64);
  GSFM_HIP_CHECK(hipMemsetAsync(xw.ix.get(), 0xff, (size_t)nslots * sizeof(int2), s));   // (-1, -1): padding
  GSFM_HIP_CHECK(hipMemsetAsync(xw.src.get(), 0xff, (size_t)nslots * sizeof(int), s));
  GSFM_HIP_CHECK(hipMemsetAsync(xw.out.get(), 0xff, (size_t)nslots * sizeof(int), s));
  // camera of every camera-major entry: the sorted keys of build_obs_graph (ws.keys_sorted)
  hipLaunchKernelGGL(k_ox_fill, dim3(grid_for(Mu, kBlock)), dim3(kBlock), 0, s, Mu, (const int*)xw.key_sorted.get(),
                     (const int*)xw.src_sorted.get(), (const int*)xw.choff.get(), (const int*)xw.xbase.get(), g.c_pt,
                     (const int*)ws.keys_sorted.get(), xw.ix.get(), xw.src.get(), xw.c_xslot.get());
  // pieces in (camera, slot) order: a stable sort of the slots by (camera of a piece tail | N)
  xw.tkey.ensure(nslots + 1);
  xw.tkey_sorted.ensure(nslots + 1);
  xw.tval.ensure(nslots + 1);
  xw.tval_sorted.ensure(nslots + 1);
  xw.piece_off.ensure(g.N + 3);
  hipLaunchKernelGGL(k_ox_tails, dim3(grid_for(nslots, kBlock)), dim3(kBlock), 0, s, nslots, g.N, (const int2*)xw.ix.get(),
                     xw.tkey.get(), xw.tval.get());
  int cbits = 1;
  while ((1L << cbits) <= g.N) ++cbits;
  sort_pairs_i32(ctx, ws.sort_tmp, xw.tkey.get(), xw.tkey_sorted.get(), xw.tval.get(), xw.tval_sorted.get(), (size_t)nslots, cbits);
  hipLaunchKernelGGL(k_og_offsets, dim3(grid_for(nslots + 1, kBlock)), dim3(kBlock), 0, s, g.N, nslots, xw.tkey_sorted.get(),
                     xw.piece_off.get());
  int* h_np = reinterpret_cast<int*>(ctx->h_pinned + 512);
  GSFM_HIP_CHECK(hipMemcpyAsync(h_np, xw.piece_off.get() + g.N, sizeof(int), hipMemcpyDeviceToHost, s));
  GSFM_HIP_CHECK(hipStreamSynchronize(s));  // (the host tables above go out of scope as well)
  const int npieces = h_np[0];
  hipLaunchKernelGGL(k_ox_out, dim3(grid_for(std::max(1, npieces), kBlock)), dim3(kBlock), 0, s, (long)npieces,
                     (const int*)xw.tval_sorted.get(), xw.out.get());
  x.tiles = (int)(nslots / 64);
  x.per = (x.tiles + 7) / 8;
  x.npieces = npieces;
  x.ix = xw.ix.get();
  x.src = xw.src.get();
  x.out = xw.out.get();
  x.piece_off = xw.piece_off.get();
  x.c_xslot = xw.c_xslot.get();
  return true;
}

// Camera-major slots of camera n (host).
inline std::vector<int> cam_slots(const ObsGraphWs& ws, int n) {
  std::vector<int> out;
  for (int k = ws.h_coff[n]; k < ws.h_coff[n + 1]; ++k) out.push_back(k);
  return out;
}

}  // namespace gsfm
