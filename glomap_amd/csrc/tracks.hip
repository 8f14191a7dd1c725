// tracks.hip — the producers of the GP / BA inputs on MI355X (gfx950):
//
//   TrackEngine::EstablishFullTracks            glomap/controllers/track_establishment.cc:5-152
//   TrackEngine::FindTracksForProblem           glomap/controllers/track_establishment.cc:154-227
//   ViewGraph::KeepLargestConnectedComponents   glomap/scene/view_graph.cc:56-97
//
// The reference walks hash maps and a pointer-chasing union-find (colmap::UnionFind over 64-bit global feature
// ids).  Here every feature is a dense 32-bit node (feat_offset[image] + feature), and the work is a handful of
// HBM-bound integer sweeps:
//
//   establish:  lock-free union-find over the inlier matches (hook the larger root under the smaller with one
//               atomicCAS, path halving on the way up => the root of a component is its SMALLEST node, i.e. the
//               canonical track id), one flatten sweep, a stream compaction of the matched features, ONE stable
//               radix sort by root (members arrive grouped by track, ascending inside it), a same-image
//               consistency sweep (one lane per member), a second compaction that drops the discarded tracks'
//               members.
//   select:     per-observation registered / first-of-its-image flags, the (length, id)-descending order as two
//               stable radix sorts, and the reference's greedy per-camera counters in closed form: a counter is
//               min(limit + 1, #earlier observations of that camera), independent of which tracks were inserted,
//               so "track s is inserted" <=> one of its observations has rank <= limit in its camera's list —
//               one more stable sort by camera instead of a sequential loop over all tracks.
//   keep-largest-component: the same union-find over the valid view-graph edges.
//
// Integer work only: results are bit-exact against oracle/tracks.py.  rocPRIM's radix sort / scan are plumbing.
#include <algorithm>

#include "common.hpp"

namespace gsfm {

void sort_pairs_i32(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const int* keys_in, int* keys_out, const int* vals_in,
                    int* vals_out, size_t n, int key_bits);
void sort_pairs_desc_u64(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const unsigned long long* keys_in,
                         unsigned long long* keys_out, const int* vals_in, int* vals_out, size_t n);
void sort_pairs_desc_u32(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const unsigned* keys_in, unsigned* keys_out,
                         const int* vals_in, int* vals_out, size_t n);
void exclusive_scan_i64(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const long* in, long* out, size_t n);

namespace {

using u64 = unsigned long long;
constexpr int kCounterStripes = 64;

// ------------------------------------------------------------------------------------------------------
// lock-free union-find (parent[x] <= x at all times; only roots are ever CAS-ed)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ld_par(const int* p, long i) {
  return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_par(int* p, long i, int v) {
  __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// representative of v with path halving.  A stale read can only return an ANCESTOR of v that is no longer a
// root; the CAS in uf_union then fails and walks on, so stale data costs time, never correctness.
__device__ __forceinline__ int uf_rep(int* parent, int v) {
  int curr = ld_par(parent, v);
  if (curr != v) {
    int prev = v, next;
    while (curr > (next = ld_par(parent, curr))) {
      st_par(parent, prev, next);
      prev = curr;
      curr = next;
    }
  }
  return curr;
}

__device__ __forceinline__ void uf_union(int* parent, int u, int v) {
  int a = uf_rep(parent, u), b = uf_rep(parent, v);
  while (a != b) {
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicCAS(parent + a, a, b);  // hook the larger root under the smaller
    if (old == a) break;
    a = old;  // a was hooked by someone else meanwhile: go on from its new parent
  }
}

__global__ void __launch_bounds__(kBlock) k_iota(long n, int* __restrict__ p) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = (int)i;
}

// LPP lanes per image pair stride its inlier matches (track_establishment.cc:19-63).  err: a feature index
// outside its image's range.
template <int LPP>
__global__ void __launch_bounds__(kBlock)
    k_uf_hook(long npairs, const int* __restrict__ p1, const int* __restrict__ p2, const unsigned char* __restrict__ pvalid,
              const long* __restrict__ poff, const unsigned* __restrict__ f1, const unsigned* __restrict__ f2,
              const long* __restrict__ foff, int num_images, int* parent, int* __restrict__ err) {
  const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub = threadIdx.x & (LPP - 1);
  const long stride = (long)gridDim.x * blockDim.x / LPP;
  for (long p = gt / LPP; p < npairs; p += stride) {
    if (pvalid && !pvalid[p]) continue;
    const int i1 = p1[p], i2 = p2[p];
    if ((unsigned)i1 >= (unsigned)num_images || (unsigned)i2 >= (unsigned)num_images) {
      *err = 1;
      continue;
    }
    const long o1 = foff[i1], n1 = foff[i1 + 1] - o1, o2 = foff[i2], n2 = foff[i2 + 1] - o2;
    const long e = poff[p + 1];
    for (long m = poff[p] + sub; m < e; m += LPP) {
      const long a = f1[m], b = f2[m];
      if (a >= n1 || b >= n2) {
        *err = 1;
        continue;
      }
      uf_union(parent, (int)(o1 + a), (int)(o2 + b));
    }
  }
}

// parent[v] = root(v); rootflag[root] = 1 for every root that has at least one other member.
__global__ void __launch_bounds__(kBlock) k_uf_flatten(long n, int* parent, int* __restrict__ rootflag) {
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) {
    int r = (int)v, p = ld_par(parent, v);
    while (p != r) {
      r = p;
      p = ld_par(parent, r);
    }
    if (r != (int)v) {
      st_par(parent, v, r);
      rootflag[r] = 1;
    }
  }
}

// a feature is a track member iff it was matched at all, i.e. iff its component has >= 2 nodes
__global__ void __launch_bounds__(kBlock)
    k_member_flag(long n, const int* __restrict__ parent, const int* __restrict__ rootflag, long* __restrict__ flag) {
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += (long)gridDim.x * blockDim.x)
    flag[v] = (v < n && (parent[v] != (int)v || rootflag[v])) ? 1 : 0;
}

__global__ void __launch_bounds__(kBlock)
    k_member_scatter(long n, const int* __restrict__ parent, const long* __restrict__ flag, const long* __restrict__ pos,
                     int* __restrict__ keys, int* __restrict__ vals) {
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x)
    if (flag[v]) {
      keys[pos[v]] = parent[v];
      vals[pos[v]] = (int)v;
    }
}

__global__ void __launch_bounds__(kBlock) k_heads(long n, const int* __restrict__ skeys, long* __restrict__ flag) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += (long)gridDim.x * blockDim.x)
    flag[k] = (k < n && (k == 0 || skeys[k] != skeys[k - 1])) ? 1 : 0;
}

__device__ __forceinline__ int image_of(const long* __restrict__ foff, int num_images, long v) {
  int lo = 0, hi = num_images;  // foff[lo] <= v < foff[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (foff[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(kBlock)
    k_decode(long n, long T, const int* __restrict__ svals, const long* __restrict__ head, const long* __restrict__ hs,
             const long* __restrict__ foff, int num_images, int* __restrict__ obs_track, int* __restrict__ obs_img,
             unsigned* __restrict__ obs_feat, long* __restrict__ tstart, long* __restrict__ tid) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const long t = hs[k] + head[k] - 1;
    const long v = svals[k];
    const int img = image_of(foff, num_images, v);
    const unsigned ft = (unsigned)(v - foff[img]);
    obs_track[k] = (int)t;
    obs_img[k] = img;
    obs_feat[k] = ft;
    if (head[k]) {
      tstart[t] = k;
      tid[t] = ((long)img << 32) | (long)ft;
    }
    if (k == n - 1) tstart[T] = n;
  }
}

// track_establishment.cc:126-137: a track dies when two of its members in ONE image are further apart than the
// threshold.  Members are ascending (image, feature), so the same-image members precede k contiguously.
__global__ void __launch_bounds__(kBlock)
    k_inconsistent(long n, const int* __restrict__ obs_track, const int* __restrict__ obs_img, const int* __restrict__ svals,
                   const long* __restrict__ tstart, const double* __restrict__ xy, double thres, int* bad) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const int t = obs_track[k], img = obs_img[k];
    const long s = tstart[t];
    if (k == s || obs_img[k - 1] != img) continue;
    const double x = xy[2 * (long)svals[k]], y = xy[2 * (long)svals[k] + 1];
    for (long j = k - 1; j >= s && obs_img[j] == img; --j) {
      if (__hip_atomic_load(bad + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      const double dx = xy[2 * (long)svals[j]] - x, dy = xy[2 * (long)svals[j] + 1] - y;
      if (__dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))) > thres) {  // Eigen norm(), no contraction
        bad[t] = 1;
        break;
      }
    }
  }
}

__global__ void __launch_bounds__(kBlock)
    k_track_len(long T, const long* __restrict__ tstart, const int* __restrict__ bad, long* __restrict__ flag, u64* counter) {
  int local = 0;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t <= T; t += (long)gridDim.x * blockDim.x) {
    const bool b = t < T && bad[t];
    flag[t] = (t < T && !b) ? tstart[t + 1] - tstart[t] : 0;
    local += b;
  }
  // same-address atomics serialise at ~15 ns each: one per wave, striped over kCounterStripes addresses
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if (local && (threadIdx.x & 63) == 0) atomicAdd(counter + (blockIdx.x % kCounterStripes), (u64)local);
}

__global__ void __launch_bounds__(kBlock)
    k_obs_final(long n, const int* __restrict__ obs_track, const int* __restrict__ bad, const long* __restrict__ tstart,
                const long* __restrict__ off, const int* __restrict__ obs_img, const unsigned* __restrict__ obs_feat,
                int* __restrict__ out_img, unsigned* __restrict__ out_feat) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const int t = obs_track[k];
    if (bad[t]) continue;
    const long d = off[t] + (k - tstart[t]);
    out_img[d] = obs_img[k];
    out_feat[d] = obs_feat[k];
  }
}

// ------------------------------------------------------------------------------------------------------
// selection
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_fill_obs_track(long T, const long* __restrict__ off, int* __restrict__ obs_track) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (long)gridDim.x * blockDim.x)
    for (long k = off[t]; k < off[t + 1]; ++k) obs_track[k] = (int)t;
}

// per observation (:188-198): bit 0 = its image is registered, bit 1 = and it is the first observation of that image
// inside the track; rrank = number of registered observations before it in the track (its slot in track_temp).
// Tracks that fail the length filter (:161-164) are skipped, which also bounds the look-back loop by
// max_num_view_per_track.
__global__ void __launch_bounds__(kBlock)
    k_sel_obs(long n, const int* __restrict__ obs_track, const long* __restrict__ off, const int* __restrict__ obs_img,
              const unsigned char* __restrict__ reg, int num_images, u64 minv, u64 maxv, unsigned char* __restrict__ oflag,
              int* __restrict__ rrank, int* __restrict__ err) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const int t = obs_track[k];
    const long s = off[t];
    const u64 len = (u64)(off[t + 1] - s);
    unsigned char f = 0;
    int rr = 0;
    if (!(len < minv || len > maxv)) {
      const int img = obs_img[k];
      if ((unsigned)img >= (unsigned)num_images) {
        *err = 1;
      } else {
        bool first = true;
        for (long j = s; j < k; ++j) {
          const int ij = obs_img[j];
          if ((unsigned)ij < (unsigned)num_images) rr += reg[ij];
          first = first && ij != img;
        }
        if (reg[img]) f = first ? 3 : 1;
      }
    }
    oflag[k] = f;
    rrank[k] = rr;
  }
}

// :161-164, :197: length window on the full track, then at least min_num_view distinct registered images
__global__ void __launch_bounds__(kBlock)
    k_sel_survive(long T, const long* __restrict__ off, const unsigned char* __restrict__ oflag, u64 minv, u64 maxv,
                  int* __restrict__ nreg, long* __restrict__ flag) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t <= T; t += (long)gridDim.x * blockDim.x) {
    long f = 0;
    if (t < T) {
      const u64 len = (u64)(off[t + 1] - off[t]);
      int nr = 0, uq = 0;
      if (!(len < minv) && !(len > maxv)) {
        for (long k = off[t]; k < off[t + 1]; ++k) {
          nr += oflag[k] & 1;
          uq += oflag[k] >> 1;
        }
        f = !((u64)uq < minv);
      }
      nreg[t] = nr;
    }
    flag[t] = f;
  }
}

__global__ void __launch_bounds__(kBlock)
    k_sel_compact(long T, const long* __restrict__ flag, const long* __restrict__ pos, const long* __restrict__ id,
                  u64* __restrict__ keys, int* __restrict__ vals) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (long)gridDim.x * blockDim.x)
    if (flag[t]) {
      keys[pos[t]] = (u64)id[t];
      vals[pos[t]] = (int)t;
    }
}

__global__ void __launch_bounds__(kBlock)
    k_len_keys(long n, const int* __restrict__ trk, const long* __restrict__ off, unsigned* __restrict__ keys) {
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (long)gridDim.x * blockDim.x)
    keys[c] = (unsigned)(off[trk[c] + 1] - off[trk[c]]);
}

__global__ void __launch_bounds__(kBlock)
    k_gather_count(long n, const int* __restrict__ order, const int* __restrict__ cnt, const unsigned char* __restrict__ mask,
                   long* __restrict__ flag) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s <= n; s += (long)gridDim.x * blockDim.x)
    flag[s] = (s < n && (!mask || mask[s])) ? (cnt ? cnt[order[s]] : 1) : 0;
}

__global__ void __launch_bounds__(kBlock) k_inv_order(long n, const int* __restrict__ order, int* __restrict__ spos) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (long)gridDim.x * blockDim.x) spos[order[s]] = (int)s;
}

// registered observations of the surviving tracks, laid out in selection order: (camera, sorted track index)
__global__ void __launch_bounds__(kBlock)
    k_fill_cam(long n, const int* __restrict__ obs_track, const int* __restrict__ spos, const long* __restrict__ base,
               const int* __restrict__ obs_img, const unsigned char* __restrict__ oflag, const int* __restrict__ rrank,
               int* __restrict__ cam_key, int* __restrict__ cam_val) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const int s = spos[obs_track[k]];
    if (s < 0 || !(oflag[k] & 1)) continue;
    const long d = base[s] + rrank[k];
    cam_key[d] = obs_img[k];
    cam_val[d] = s;
  }
}

__global__ void __launch_bounds__(kBlock) k_cam_heads(long n, const int* __restrict__ skey, long* __restrict__ camstart) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x)
    if (k == 0 || skey[k] != skey[k - 1]) camstart[skey[k]] = k;
}

// :203-216 in closed form: the counter an observation sees is min(limit + 1, its rank in the camera's list)
__global__ void __launch_bounds__(kBlock)
    k_hit(long n, const int* __restrict__ skey, const int* __restrict__ sval, const long* __restrict__ camstart, u64 limit,
          unsigned char* __restrict__ added) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x)
    if ((u64)(k - camstart[skey[k]]) <= limit) added[sval[k]] = 1;
}

// :221-222: the loop stops once tracks.size() > max_num_tracks, i.e. a track is inserted iff fewer than or exactly
// max_num_tracks tracks were inserted before it
__global__ void __launch_bounds__(kBlock)
    k_chosen(long n, const unsigned char* __restrict__ added, const long* __restrict__ pre, u64 max_tracks,
             unsigned char* __restrict__ chosen) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (long)gridDim.x * blockDim.x)
    chosen[s] = added[s] && (u64)pre[s] <= max_tracks;
}

__global__ void __launch_bounds__(kBlock)
    k_sel_emit_tracks(long n, long S, long Ro, const unsigned char* __restrict__ chosen, const long* __restrict__ slot,
                      const long* __restrict__ obase, const int* __restrict__ order, const long* __restrict__ id,
                      long* __restrict__ out_id, long* __restrict__ out_off) {
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (long)gridDim.x * blockDim.x) {
    if (s == 0) out_off[S] = Ro;
    if (!chosen[s]) continue;
    out_id[slot[s]] = id[order[s]];
    out_off[slot[s]] = obase[s];
  }
}

__global__ void __launch_bounds__(kBlock)
    k_sel_emit_obs(long n, const int* __restrict__ obs_track, const int* __restrict__ spos, const unsigned char* __restrict__ chosen,
                   const long* __restrict__ obase, const int* __restrict__ obs_img, const unsigned* __restrict__ obs_feat,
                   const unsigned char* __restrict__ oflag, const int* __restrict__ rrank, int* __restrict__ out_img,
                   unsigned* __restrict__ out_feat) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const int s = spos[obs_track[k]];
    if (s < 0 || !chosen[s] || !(oflag[k] & 1)) continue;
    const long d = obase[s] + rrank[k];
    out_img[d] = obs_img[k];
    out_feat[d] = obs_feat[k];
  }
}

// ------------------------------------------------------------------------------------------------------
// largest connected component of the view graph
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_cc_hook(long E, int N, const int* __restrict__ ei, const int* __restrict__ ej, const unsigned char* __restrict__ valid,
              int* parent, int* __restrict__ touched, int* __restrict__ err) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
    if (!valid[e]) continue;
    const int i = ei[e], j = ej[e];
    if ((unsigned)i >= (unsigned)N || (unsigned)j >= (unsigned)N) {
      *err = 1;
      continue;
    }
    touched[i] = 1;
    touched[j] = 1;
    uf_union(parent, i, j);
  }
}

__global__ void __launch_bounds__(kBlock)
    k_cc_count(int N, const int* __restrict__ parent, const int* __restrict__ touched, int* cnt) {
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x)
    if (touched[v]) atomicAdd(cnt + parent[v], 1);
}

// argmax of (size, -root): the largest component, ties to the one holding the smallest node.  One block.
__global__ void __launch_bounds__(1024) k_cc_best(int N, const int* __restrict__ cnt, u64* __restrict__ best) {
  __shared__ u64 sh[1024];
  u64 b = 0;
  for (int v = threadIdx.x; v < N; v += blockDim.x)
    if (cnt[v] > 0) {
      const u64 key = ((u64)cnt[v] << 32) | (u64)(0x7fffffff - v);
      b = key > b ? key : b;
    }
  sh[threadIdx.x] = b;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] > sh[threadIdx.x + s] ? sh[threadIdx.x] : sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *best = sh[0];
}

__global__ void __launch_bounds__(kBlock)
    k_cc_nodes(int N, const int* __restrict__ parent, const int* __restrict__ touched, const u64* __restrict__ best,
               const int* __restrict__ nimg, unsigned char* __restrict__ reg, u64* counter) {
  const int root = 0x7fffffff - (int)(*best & 0xffffffffu);
  int local = 0;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
    const bool r = touched[v] && parent[v] == root;
    reg[v] = r;
    if (r) local += nimg ? nimg[v] : 1;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if (local && (threadIdx.x & 63) == 0) atomicAdd(counter, (u64)local);
}

__global__ void __launch_bounds__(kBlock)
    k_cc_edges(long E, const int* __restrict__ ei, const int* __restrict__ ej, const unsigned char* __restrict__ reg,
               unsigned char* __restrict__ valid) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x)
    if (!reg[ei[e]] || !reg[ej[e]]) valid[e] = 0;  // view_graph.cc:85-90
}

// ------------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------------
struct TrackSetDev {
  long T = 0, Mo = 0;
  DevBuf<long> id, off;
  DevBuf<int> img;
  DevBuf<unsigned> feat;
};

struct TracksWs {
  // staged inputs
  DevBuf<long> foff, poff;
  DevBuf<double> xy;
  DevBuf<int> p1, p2;
  DevBuf<unsigned char> pvalid, reg;
  DevBuf<unsigned> f1, f2;
  DevBuf<long> in_id, in_off;
  DevBuf<int> in_img;
  DevBuf<unsigned> in_feat;
  // scratch
  DevBuf<int> parent, rootflag, keys, vals, skeys, svals, obs_track, obs_img, bad, nreg, uniq, err;
  DevBuf<unsigned> obs_feat, k32, sk32;
  DevBuf<long> flag, scan, scan2, tstart, camstart;
  DevBuf<u64> k64, sk64, counter;
  DevBuf<unsigned char> sort_tmp, added, chosen, oflag;
  DevBuf<int> rrank;
  TrackSetDev full, sel;
  static void destroy(void* p) { delete static_cast<TracksWs*>(p); }
};

TracksWs* tracks_ws(gsfm_ctx* ctx) {
  if (!ctx->tr_ws) {
    ctx->tr_ws = new TracksWs();
    ctx->tr_ws_free = &TracksWs::destroy;
  }
  return static_cast<TracksWs*>(ctx->tr_ws);
}

template <typename T>
const T* dev_in(gsfm_ctx* ctx, DevBuf<T>& buf, const T* src, size_t n, int mem) {
  if (mem == GSFM_MEM_DEVICE) return src;
  T* d = buf.ensure(n + 1);
  copy_in(ctx, d, src, n, mem);
  return d;
}

long read_long(gsfm_ctx* ctx, const long* dev) {
  long* h = reinterpret_cast<long*>(ctx->h_pinned);
  GSFM_HIP_CHECK(hipMemcpyAsync(h, dev, sizeof(long), hipMemcpyDeviceToHost, ctx->stream));
  GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return h[0];
}

int bits_for(long n) {
  int b = 1;
  while ((1L << b) < n && b < 31) ++b;
  return b;
}

inline u64 as_u64(int32_t x) { return (u64)(long)x; }  // C++'s int -> size_t conversion

#define LAUNCH(kernel, work, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid_wide((size_t)(work), kBlock, 1 << 16)), dim3(kBlock), 0, s, __VA_ARGS__)

int establish_impl(gsfm_ctx* ctx, const gsfm_match_graph* g, const gsfm_track_options* opt, int64_t* num_tracks,
                   int64_t* num_obs, int64_t* num_discarded) {
  GSFM_REQUIRE(g && opt, "tracks: null argument");
  GSFM_REQUIRE(g->num_images > 0 && g->feat_offset && g->num_pairs >= 0, "tracks: bad match graph");
  GSFM_REQUIRE(g->num_pairs == 0 || (g->pair_image1 && g->pair_image2 && g->pair_offset), "tracks: null pair arrays");
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  TracksWs* ws = tracks_ws(ctx);
  hipStream_t s = ctx->stream;
  const int I = g->num_images;
  const long NP = g->num_pairs;
  const long* foff = dev_in(ctx, ws->foff, reinterpret_cast<const long*>(g->feat_offset), (size_t)I + 1, g->mem);
  std::vector<long> ends;
  to_host(ctx, ends, reinterpret_cast<const long*>(g->feat_offset) + I, 1, g->mem);
  const long F = ends[0];
  long NM = 0;
  if (NP > 0) {
    to_host(ctx, ends, reinterpret_cast<const long*>(g->pair_offset) + NP, 1, g->mem);
    NM = ends[0];
  }
  if (F >= (1L << 31)) throw StatusError(GSFM_ERR_UNSUPPORTED, "tracks: more than 2^31 features");
  GSFM_REQUIRE(F >= 0 && NM >= 0 && (NM == 0 || (g->match_feat1 && g->match_feat2 && g->feat_xy)), "tracks: null match arrays");
  TrackSetDev& out = ws->full;
  out.T = out.Mo = 0;
  long discarded = 0;
  if (F > 0 && NM > 0) {
    const int* p1 = dev_in(ctx, ws->p1, g->pair_image1, (size_t)NP, g->mem);
    const int* p2 = dev_in(ctx, ws->p2, g->pair_image2, (size_t)NP, g->mem);
    const unsigned char* pv = g->pair_valid ? dev_in(ctx, ws->pvalid, g->pair_valid, (size_t)NP, g->mem) : nullptr;
    const long* poff = dev_in(ctx, ws->poff, reinterpret_cast<const long*>(g->pair_offset), (size_t)NP + 1, g->mem);
    const unsigned* f1 = dev_in(ctx, ws->f1, g->match_feat1, (size_t)NM, g->mem);
    const unsigned* f2 = dev_in(ctx, ws->f2, g->match_feat2, (size_t)NM, g->mem);
    const double* xy = dev_in(ctx, ws->xy, g->feat_xy, 2 * (size_t)F, g->mem);
    int* parent = ws->parent.ensure(F + 1);
    int* rootflag = ws->rootflag.ensure(F + 1);
    int* err = ws->err.ensure(1);
    GSFM_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), s));
    GSFM_HIP_CHECK(hipMemsetAsync(rootflag, 0, (size_t)F * sizeof(int), s));
    LAUNCH(k_iota, F, F, parent);
    {
      const long avg = NM / std::max<long>(NP, 1);
      const bool timed = ctx->prof.begin(s, GSFM_KERNEL_TRACK_HOOK);
      // >= ~3 matches per lane keeps the lanes busy while their neighbours spin in the CAS loop.  Measured on
      // C3-scale input (5.7 M matches, 23 per pair): 4 lanes 428 us, 16 lanes 508 us, 64 lanes 527 us; the same
      // kernel with the CAS replaced by a plain store runs in 108 us, i.e. the sweep is bound by the device's
      // returning-atomic rate (~10 G CAS/s), not by HBM or by the pointer chasing.
      if (avg >= 192)
        LAUNCH(k_uf_hook<64>, NP * 64, NP, p1, p2, pv, poff, f1, f2, foff, I, parent, err);
      else if (avg >= 48)
        LAUNCH(k_uf_hook<16>, NP * 16, NP, p1, p2, pv, poff, f1, f2, foff, I, parent, err);
      else
        LAUNCH(k_uf_hook<4>, NP * 4, NP, p1, p2, pv, poff, f1, f2, foff, I, parent, err);
      if (timed) ctx->prof.end(s);
    }
    LAUNCH(k_uf_flatten, F, F, parent, rootflag);
    long* flag = ws->flag.ensure(F + 2);
    long* scan = ws->scan.ensure(F + 2);
    LAUNCH(k_member_flag, F + 1, F, parent, rootflag, flag);
    exclusive_scan_i64(ctx, ws->sort_tmp, flag, scan, (size_t)F + 1);
    const long Tm = read_long(ctx, scan + F);
    int h_err = 0;
    GSFM_HIP_CHECK(hipMemcpy(&h_err, err, sizeof(int), hipMemcpyDeviceToHost));
    GSFM_REQUIRE(h_err == 0, "tracks: image or feature index out of range");
    if (Tm > 0) {
      int* keys = ws->keys.ensure(Tm + 1);
      int* vals = ws->vals.ensure(Tm + 1);
      int* skeys = ws->skeys.ensure(Tm + 1);
      int* svals = ws->svals.ensure(Tm + 1);
      LAUNCH(k_member_scatter, F, F, parent, flag, scan, keys, vals);
      sort_pairs_i32(ctx, ws->sort_tmp, keys, skeys, vals, svals, (size_t)Tm, bits_for(F));
      LAUNCH(k_heads, Tm + 1, Tm, skeys, flag);
      exclusive_scan_i64(ctx, ws->sort_tmp, flag, scan, (size_t)Tm + 1);
      const long T = read_long(ctx, scan + Tm);
      int* obs_track = ws->obs_track.ensure(Tm + 1);
      int* obs_img = ws->obs_img.ensure(Tm + 1);
      unsigned* obs_feat = ws->obs_feat.ensure(Tm + 1);
      long* tstart = ws->tstart.ensure(T + 2);
      long* tid = out.id.ensure(T + 1);
      int* bad = ws->bad.ensure(T + 1);
      u64* counter = ws->counter.ensure(kCounterStripes);
      GSFM_HIP_CHECK(hipMemsetAsync(bad, 0, (size_t)T * sizeof(int), s));
      GSFM_HIP_CHECK(hipMemsetAsync(counter, 0, kCounterStripes * sizeof(u64), s));
      LAUNCH(k_decode, Tm, Tm, T, svals, flag, scan, foff, I, obs_track, obs_img, obs_feat, tstart, tid);
      LAUNCH(k_inconsistent, Tm, Tm, obs_track, obs_img, svals, tstart, xy, opt->thres_inconsistency, bad);
      long* lens = ws->scan2.ensure(T + 2);
      long* off = out.off.ensure(T + 2);
      LAUNCH(k_track_len, T + 1, T, tstart, bad, lens, counter);
      exclusive_scan_i64(ctx, ws->sort_tmp, lens, off, (size_t)T + 1);
      long* h = reinterpret_cast<long*>(ctx->h_pinned);
      GSFM_HIP_CHECK(hipMemcpyAsync(h, off + T, sizeof(long), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipMemcpyAsync(h + 1, counter, kCounterStripes * sizeof(u64), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipStreamSynchronize(s));
      const long Mo = h[0];
      for (int i = 0; i < kCounterStripes; ++i) discarded += h[1 + i];
      int* oimg = out.img.ensure(Mo + 1);
      unsigned* ofeat = out.feat.ensure(Mo + 1);
      LAUNCH(k_obs_final, Tm, Tm, obs_track, bad, tstart, off, obs_img, obs_feat, oimg, ofeat);
      out.T = T;
      out.Mo = Mo;
    }
  }
  if (out.T == 0) {
    long zero = 0;
    GSFM_HIP_CHECK(hipMemcpyAsync(out.off.ensure(2), &zero, sizeof(long), hipMemcpyHostToDevice, s));
  }
  GSFM_HIP_CHECK(hipStreamSynchronize(s));
  GSFM_HIP_CHECK(hipGetLastError());
  if (num_tracks) *num_tracks = out.T;
  if (num_obs) *num_obs = out.Mo;
  if (num_discarded) *num_discarded = discarded;
  return GSFM_OK;
}

int select_impl(gsfm_ctx* ctx, const gsfm_track_set* full, int32_t num_images, const uint8_t* image_registered, int32_t mem,
                const gsfm_track_options* opt, int64_t* num_tracks, int64_t* num_obs) {
  GSFM_REQUIRE(opt && image_registered && num_images > 0, "tracks: null argument");
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  TracksWs* ws = tracks_ws(ctx);
  hipStream_t s = ctx->stream;
  long T, Mo;
  const long *id, *off;
  const int* oimg;
  const unsigned* ofeat;
  if (full) {
    T = full->num_tracks;
    Mo = full->num_obs;
    GSFM_REQUIRE(T >= 0 && Mo >= 0 && (T == 0 || (full->track_id && full->track_offset)) && (Mo == 0 || (full->obs_image && full->obs_feature)),
                 "tracks: bad track set");
    GSFM_REQUIRE(T < (1L << 31), "tracks: more than 2^31 tracks");
    id = dev_in(ctx, ws->in_id, reinterpret_cast<const long*>(full->track_id), (size_t)T, full->mem);
    off = dev_in(ctx, ws->in_off, reinterpret_cast<const long*>(full->track_offset), (size_t)T + (T > 0), full->mem);
    oimg = dev_in(ctx, ws->in_img, full->obs_image, (size_t)Mo, full->mem);
    ofeat = dev_in(ctx, ws->in_feat, full->obs_feature, (size_t)Mo, full->mem);
  } else {
    T = ws->full.T;
    Mo = ws->full.Mo;
    id = ws->full.id.get();
    off = ws->full.off.get();
    oimg = ws->full.img.get();
    ofeat = ws->full.feat.get();
  }
  const unsigned char* reg = dev_in(ctx, ws->reg, image_registered, (size_t)num_images, mem);
  TrackSetDev& out = ws->sel;
  out.T = out.Mo = 0;
  const u64 minv = as_u64(opt->min_num_view_per_track), maxv = as_u64(opt->max_num_view_per_track);
  const u64 limit = as_u64(opt->min_num_tracks_per_view), max_tracks = as_u64(opt->max_num_tracks);
  long Ns = 0;
  if (T > 0) {
    int* obs_track = ws->obs_track.ensure(Mo + 1);
    int* nreg = ws->nreg.ensure(T + 1);
    int* spos = ws->uniq.ensure(T + 1);
    int* rrank = ws->rrank.ensure(Mo + 1);
    unsigned char* oflag = ws->oflag.ensure(Mo + 1);
    int* err = ws->err.ensure(1);
    GSFM_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), s));
    GSFM_HIP_CHECK(hipMemsetAsync(spos, 0xFF, (size_t)T * sizeof(int), s));
    LAUNCH(k_fill_obs_track, T, T, off, obs_track);
    if (Mo > 0) LAUNCH(k_sel_obs, Mo, Mo, obs_track, off, oimg, reg, num_images, minv, maxv, oflag, rrank, err);
    long* flag = ws->flag.ensure(std::max(T, Mo) + 2);
    long* scan = ws->scan.ensure(std::max(T, Mo) + 2);
    LAUNCH(k_sel_survive, T + 1, T, off, oflag, minv, maxv, nreg, flag);
    exclusive_scan_i64(ctx, ws->sort_tmp, flag, scan, (size_t)T + 1);
    Ns = read_long(ctx, scan + T);
    int h_err = 0;
    GSFM_HIP_CHECK(hipMemcpy(&h_err, err, sizeof(int), hipMemcpyDeviceToHost));
    GSFM_REQUIRE(h_err == 0, "tracks: image index out of range");
    if (Ns > 0) {
      // (length, id) descending = stable sort by id, then stable sort by length (:167)
      u64* k64 = ws->k64.ensure(Ns + 1);
      u64* sk64 = ws->sk64.ensure(Ns + 1);
      int* vals = ws->vals.ensure(Ns + 1);
      int* svals = ws->svals.ensure(Ns + 1);
      int* order = ws->keys.ensure(Ns + 1);
      unsigned* k32 = ws->k32.ensure(Ns + 1);
      unsigned* sk32 = ws->sk32.ensure(Ns + 1);
      LAUNCH(k_sel_compact, T, T, flag, scan, id, k64, vals);
      sort_pairs_desc_u64(ctx, ws->sort_tmp, k64, sk64, vals, svals, (size_t)Ns);
      LAUNCH(k_len_keys, Ns, Ns, svals, off, k32);
      sort_pairs_desc_u32(ctx, ws->sort_tmp, k32, sk32, svals, order, (size_t)Ns);
      LAUNCH(k_inv_order, Ns, Ns, order, spos);
      unsigned char* added = ws->added.ensure(Ns + 1);
      unsigned char* chosen = ws->chosen.ensure(Ns + 1);
      if (limit >= (u64)Mo) {  // no camera can reach the cap: every surviving track is inserted
        GSFM_HIP_CHECK(hipMemsetAsync(added, 1, (size_t)Ns, s));
      } else {
        GSFM_HIP_CHECK(hipMemsetAsync(added, 0, (size_t)Ns, s));
        LAUNCH(k_gather_count, Ns + 1, Ns, order, nreg, (const unsigned char*)nullptr, flag);
        exclusive_scan_i64(ctx, ws->sort_tmp, flag, scan, (size_t)Ns + 1);
        const long R = read_long(ctx, scan + Ns);
        if (R > 0) {
          int* ck = ws->obs_img.ensure(R + 1);
          int* cv = ws->bad.ensure(R + 1);
          int* sck = ws->skeys.ensure(R + 1);
          int* scv = ws->parent.ensure(R + 1);
          long* camstart = ws->camstart.ensure((size_t)num_images + 1);
          LAUNCH(k_fill_cam, Mo, Mo, obs_track, spos, scan, oimg, oflag, rrank, ck, cv);
          sort_pairs_i32(ctx, ws->sort_tmp, ck, sck, cv, scv, (size_t)R, bits_for(num_images));
          LAUNCH(k_cam_heads, R, R, sck, camstart);
          LAUNCH(k_hit, R, R, sck, scv, camstart, limit, added);
        }
      }
      long* pre = ws->scan2.ensure(Ns + 2);
      LAUNCH(k_gather_count, Ns + 1, Ns, order, (const int*)nullptr, added, flag);
      exclusive_scan_i64(ctx, ws->sort_tmp, flag, pre, (size_t)Ns + 1);
      LAUNCH(k_chosen, Ns, Ns, added, pre, max_tracks, chosen);
      long* slot = ws->tstart.ensure(Ns + 2);
      LAUNCH(k_gather_count, Ns + 1, Ns, order, (const int*)nullptr, chosen, flag);
      exclusive_scan_i64(ctx, ws->sort_tmp, flag, slot, (size_t)Ns + 1);
      const long S = read_long(ctx, slot + Ns);
      LAUNCH(k_gather_count, Ns + 1, Ns, order, nreg, chosen, flag);
      exclusive_scan_i64(ctx, ws->sort_tmp, flag, scan, (size_t)Ns + 1);
      const long Ro = read_long(ctx, scan + Ns);
      LAUNCH(k_sel_emit_tracks, Ns, Ns, S, Ro, chosen, slot, scan, order, id, out.id.ensure(S + 1), out.off.ensure(S + 2));
      if (Ro > 0)
        LAUNCH(k_sel_emit_obs, Mo, Mo, obs_track, spos, chosen, scan, oimg, ofeat, oflag, rrank, out.img.ensure(Ro + 1),
               out.feat.ensure(Ro + 1));
      out.T = S;
      out.Mo = Ro;
    }
  }
  if (out.T == 0) {
    long zero = 0;
    GSFM_HIP_CHECK(hipMemcpyAsync(out.off.ensure(2), &zero, sizeof(long), hipMemcpyHostToDevice, s));
  }
  GSFM_HIP_CHECK(hipStreamSynchronize(s));
  GSFM_HIP_CHECK(hipGetLastError());
  if (num_tracks) *num_tracks = out.T;
  if (num_obs) *num_obs = out.Mo;
  return GSFM_OK;
}

int fetch_impl(gsfm_ctx* ctx, int32_t which, gsfm_track_set* o) {
  GSFM_REQUIRE(o && (which == GSFM_TRACKS_FULL || which == GSFM_TRACKS_SELECTED), "tracks: bad fetch argument");
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  TracksWs* ws = tracks_ws(ctx);
  const TrackSetDev& src = which == GSFM_TRACKS_FULL ? ws->full : ws->sel;
  GSFM_REQUIRE(src.off.get() != nullptr, "tracks: nothing to fetch (run establish / select first)");
  GSFM_REQUIRE(o->track_offset && (src.T == 0 || o->track_id) && (src.Mo == 0 || (o->obs_image && o->obs_feature)),
               "tracks: null output array");
  copy_out(ctx, reinterpret_cast<long*>(o->track_id), src.id.get(), (size_t)src.T, o->mem);
  copy_out(ctx, reinterpret_cast<long*>(o->track_offset), src.off.get(), (size_t)src.T + 1, o->mem);
  copy_out(ctx, o->obs_image, src.img.get(), (size_t)src.Mo, o->mem);
  copy_out(ctx, o->obs_feature, src.feat.get(), (size_t)src.Mo, o->mem);
  GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  o->num_tracks = src.T;
  o->num_obs = src.Mo;
  return GSFM_OK;
}

int keep_largest_impl(gsfm_ctx* ctx, int32_t mem, int32_t N, int64_t E, const int32_t* edge_i, const int32_t* edge_j,
                      uint8_t* edge_valid, const int32_t* node_num_images, uint8_t* node_registered, int64_t* num_images_out) {
  GSFM_REQUIRE(N > 0 && E >= 0 && node_registered && (E == 0 || (edge_i && edge_j && edge_valid)), "keep-largest: null argument");
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  TracksWs* ws = tracks_ws(ctx);
  hipStream_t s = ctx->stream;
  long result = 0;
  if (E > 0) {
    const int* ei = dev_in(ctx, ws->p1, edge_i, (size_t)E, mem);
    const int* ej = dev_in(ctx, ws->p2, edge_j, (size_t)E, mem);
    unsigned char* valid = edge_valid;
    if (mem != GSFM_MEM_DEVICE) {
      valid = ws->pvalid.ensure(E + 1);
      copy_in(ctx, valid, edge_valid, (size_t)E, mem);
    }
    const int* nimg = node_num_images ? dev_in(ctx, ws->in_img, node_num_images, (size_t)N, mem) : nullptr;
    int* parent = ws->parent.ensure(N + 1);
    int* touched = ws->rootflag.ensure(N + 1);
    int* cnt = ws->nreg.ensure(N + 1);
    int* dummy = ws->uniq.ensure(N + 1);
    int* err = ws->err.ensure(1);
    u64* counter = ws->counter.ensure(2);
    unsigned char* reg = mem == GSFM_MEM_DEVICE ? node_registered : ws->reg.ensure(N + 1);
    GSFM_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), s));
    GSFM_HIP_CHECK(hipMemsetAsync(touched, 0, (size_t)N * sizeof(int), s));
    GSFM_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)N * sizeof(int), s));
    GSFM_HIP_CHECK(hipMemsetAsync(counter, 0, 2 * sizeof(u64), s));
    LAUNCH(k_iota, N, (long)N, parent);
    LAUNCH(k_cc_hook, E, (long)E, N, ei, ej, valid, parent, touched, err);
    LAUNCH(k_uf_flatten, N, (long)N, parent, dummy);
    LAUNCH(k_cc_count, N, N, parent, touched, cnt);
    hipLaunchKernelGGL(k_cc_best, dim3(1), dim3(1024), 0, s, N, cnt, counter + 1);
    u64 h[2];
    GSFM_HIP_CHECK(hipMemcpyAsync(h, counter, 2 * sizeof(u64), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    int h_err = 0;
    GSFM_HIP_CHECK(hipMemcpy(&h_err, err, sizeof(int), hipMemcpyDeviceToHost));
    GSFM_REQUIRE(h_err == 0, "keep-largest: node index out of range");
    if ((h[1] >> 32) > 0) {  // view_graph.cc:70: no component -> return 0, nothing touched
      LAUNCH(k_cc_nodes, N, N, parent, touched, counter + 1, nimg, reg, counter);
      LAUNCH(k_cc_edges, E, (long)E, ei, ej, reg, valid);
      if (mem != GSFM_MEM_DEVICE) {
        copy_out(ctx, edge_valid, valid, (size_t)E, mem);
        copy_out(ctx, node_registered, reg, (size_t)N, mem);
      }
      result = read_long(ctx, reinterpret_cast<const long*>(counter));
    }
  }
  GSFM_HIP_CHECK(hipStreamSynchronize(s));
  GSFM_HIP_CHECK(hipGetLastError());
  if (num_images_out) *num_images_out = result;
  return GSFM_OK;
}

}  // namespace
}  // namespace gsfm

using namespace gsfm;

extern "C" void gsfm_track_options_default(gsfm_track_options* o) {
  if (!o) return;
  o->thres_inconsistency = 10.0;   // track_establishment.h:11
  o->min_num_tracks_per_view = -1; // :14
  o->min_num_view_per_track = 3;   // :17
  o->max_num_view_per_track = 100; // :20
  o->max_num_tracks = 10000000;    // :23
}

extern "C" int gsfm_tracks_establish(gsfm_ctx* ctx, const gsfm_match_graph* graph, const gsfm_track_options* opt,
                                     int64_t* num_tracks, int64_t* num_obs, int64_t* num_discarded) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] { return establish_impl(ctx, graph, opt, num_tracks, num_obs, num_discarded); });
}

extern "C" int gsfm_tracks_select(gsfm_ctx* ctx, const gsfm_track_set* full, int32_t num_images, const uint8_t* image_registered,
                                  int32_t mem, const gsfm_track_options* opt, int64_t* num_tracks, int64_t* num_obs) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] { return select_impl(ctx, full, num_images, image_registered, mem, opt, num_tracks, num_obs); });
}

extern "C" int gsfm_tracks_fetch(gsfm_ctx* ctx, int32_t which, gsfm_track_set* out) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] { return fetch_impl(ctx, which, out); });
}

extern "C" int gsfm_keep_largest_connected_component(gsfm_ctx* ctx, int32_t mem, int32_t num_nodes, int64_t num_edges,
                                                     const int32_t* edge_i, const int32_t* edge_j, uint8_t* edge_valid_inout,
                                                     const int32_t* node_num_images, uint8_t* node_registered_out,
                                                     int64_t* num_images_out) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    return keep_largest_impl(ctx, mem, num_nodes, num_edges, edge_i, edge_j, edge_valid_inout, node_num_images,
                             node_registered_out, num_images_out);
  });
}
