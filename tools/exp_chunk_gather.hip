// exp_chunk_gather.hip — does an XCD-partitioned camera-major sweep turn its point-record gathers into L2 hits?
//
// tools/exp_gather_calib.hip: a random 64-byte record gather over a table that does not fit an XCD's 4 MB L2 runs at
// ~54 G line requests/s whatever the record size — exactly where k_gp_phaseB sits (6.0 M gathers in 118 us).  This
// program emulates the alternative layout before gp.hip is rebuilt around it: the observations of the camera-major order
// re-sorted by (point chunk, camera), chunk c owned by XCD c % 8 (block b runs on XCD b % 8: observed dispatch rule, used
// for speed only), each XCD walking its chunks one after the other, so that the records a block gathers are the
// <= 1-2 MB of ONE chunk, resident in that XCD's L2.  One lane per observation in 64-slot tiles (the tile carries
// (point, camera) and two coefficients as coalesced streams), 64-byte (X_p, t_p) record gather, 48-byte (c_n, z_n)
// gather from a 480 KB table, wave segmented sum over the camera key, one 24-byte partial per (camera, tile) piece.
//
//   hipcc -O3 --offload-arch=gfx950 tools/exp_chunk_gather.hip -o tools/exp_chunk_gather && tools/exp_chunk_gather
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

#define CHECK(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__);         \
      exit(1);                                                                                    \
    }                                                                                             \
  } while (0)

// tile t of the launch -> slot base; MODE 0: tiles in list order (block b takes tiles 4 b .. 4 b + 3: consecutive tiles land
// on different XCDs); MODE 1: the list is cut in 8 equal parts, block b works on part b % 8 (XCD-affine)
template <int MODE, int REC>
__global__ void __launch_bounds__(256)
    k_chunk_sweep(const int2* __restrict__ slot, const double2* __restrict__ coef, const double* __restrict__ rec,
                  const double* __restrict__ cz, long ntiles, double* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  long tile;
  if (MODE == 0) {
    tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  } else {
    const long per = (ntiles + 7) / 8;
    const long j = (long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    if (j >= per) return;
    tile = (long)(blockIdx.x & 7) * per + j;
  }
  if (tile >= ntiles) return;
  const int2 ix = slot[tile * 64 + lane];
  const double2 q = coef[tile * 64 + lane];
  double acc[3] = {0, 0, 0};
  if (ix.x >= 0) {
    const double2* r = reinterpret_cast<const double2*>(rec + (long)ix.x * (REC / 8));
    const double2* c = reinterpret_cast<const double2*>(cz + 6 * (long)ix.y);
    const double2 r0 = r[0], r1 = r[1];
    double2 r2 = make_double2(0, 0);
    if (REC >= 48) r2 = r[2];
    const double2 c0 = c[0], c1 = c[1], c2 = c[2];
    // y = a (v - beta d (d.v)),  d = X - c,  v = z - t
    const double dx = r0.x - c0.x, dy = r0.y - c0.y, dz = r1.x - c1.x;
    const double vx = c1.y - r1.y, vy = c2.x - r2.x, vz = c2.y - r2.y;
    const double dv = dx * vx + dy * vy + dz * vz;
    acc[0] = q.x * (vx - q.y * dx * dv);
    acc[1] = q.x * (vy - q.y * dy * dv);
    acc[2] = q.x * (vz - q.y * dz * dv);
  }
  // segmented inclusive scan over the camera key (lanes of one camera are consecutive)
  const int key = ix.x >= 0 ? ix.y : -1 - lane;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int ku = __shfl_up(key, d, 64);
    const bool take = lane >= d && ku == key;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double vu = __shfl_up(acc[k], d, 64);
      if (take) acc[k] += vu;
    }
  }
  const int kn = __shfl_down(key, 1, 64);
  if ((lane == 63 || kn != key) && key >= 0) {  // tail of a (camera, tile) piece: one 24-byte partial (slot = its lane: stand-in)
    double* o = part + 3 * (tile * 64 + lane);
    o[0] = acc[0];
    o[1] = acc[1];
    o[2] = acc[2];
  }
}

static float time_launches(int reps, hipStream_t s, const std::function<void()>& launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const int N = 10000;
  const long P = 1'000'000;
  const int per_cam = 600;  // observations per camera
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  std::mt19937_64 rng(1);
  // camera-major observation lists: camera n sees per_cam random points, ascending
  std::vector<std::vector<int>> lists(N);
  for (int n = 0; n < N; ++n) {
    auto& l = lists[n];
    l.resize(per_cam);
    for (auto& p : l) p = (int)(rng() % (unsigned long long)P);
    std::sort(l.begin(), l.end());
  }
  double *rec = nullptr, *cz = nullptr, *part = nullptr;
  CHECK(hipMalloc((void**)&rec, (size_t)P * 64));
  CHECK(hipMalloc((void**)&cz, (size_t)N * 48));
  CHECK(hipMemset(rec, 0, (size_t)P * 64));
  CHECK(hipMemset(cz, 0, (size_t)N * 48));
  for (int rec_bytes : {64, 32}) {
    for (int C : {1, 8, 16, 24, 32, 48, 64, 96}) {
      // order (chunk, camera, point); chunk c = points [c Pc, (c + 1) Pc); XCD x owns chunks x R .. x R + R - 1 (R = C / 8), so
      // that cutting the list in 8 equal parts gives every XCD (about) its own chunks in sequence
      const long Pc = (P + C - 1) / C;
      std::vector<std::vector<int2>> by_chunk(C);
      for (int n = 0; n < N; ++n)
        for (int p : lists[n]) by_chunk[p / Pc].push_back(make_int2(p, n));
      std::vector<int2> slots;
      long pieces = 0;
      for (int c = 0; c < C; ++c) {
        std::stable_sort(by_chunk[c].begin(), by_chunk[c].end(), [](const int2& a, const int2& b) { return a.y < b.y; });
        for (auto& v : by_chunk[c]) slots.push_back(v);
        while (slots.size() % 64) slots.push_back(make_int2(-1, 0));  // chunks start on a tile boundary
      }
      const long ntiles = (long)slots.size() / 64;
      for (long t = 0; t < ntiles; ++t)
        for (int l = 0; l < 64; ++l) {
          const int2 a = slots[t * 64 + l];
          if (a.x >= 0 && (l == 63 || slots[t * 64 + l + 1].y != a.y || slots[t * 64 + l + 1].x < 0)) ++pieces;
        }
      int2* d_slot = nullptr;
      double2* d_coef = nullptr;
      CHECK(hipMalloc((void**)&d_slot, slots.size() * sizeof(int2)));
      CHECK(hipMalloc((void**)&d_coef, slots.size() * sizeof(double2)));
      CHECK(hipMemcpy(d_slot, slots.data(), slots.size() * sizeof(int2), hipMemcpyHostToDevice));
      CHECK(hipMemset(d_coef, 0, slots.size() * sizeof(double2)));
      if (part) CHECK(hipFree(part));
      CHECK(hipMalloc((void**)&part, slots.size() * 24));
      const int grid0 = (int)((ntiles + 3) / 4);
      const long per = (ntiles + 7) / 8;
      const int grid1 = (int)(8 * ((per + 3) / 4));
      float ms0, ms1;
      if (rec_bytes == 64) {
        ms0 = time_launches(reps, s, [&] { hipLaunchKernelGGL((k_chunk_sweep<0, 64>), dim3(grid0), dim3(256), 0, s, d_slot, d_coef, rec, cz, ntiles, part); });
        ms1 = time_launches(reps, s, [&] { hipLaunchKernelGGL((k_chunk_sweep<1, 64>), dim3(grid1), dim3(256), 0, s, d_slot, d_coef, rec, cz, ntiles, part); });
      } else {
        ms0 = time_launches(reps, s, [&] { hipLaunchKernelGGL((k_chunk_sweep<0, 32>), dim3(grid0), dim3(256), 0, s, d_slot, d_coef, rec, cz, ntiles, part); });
        ms1 = time_launches(reps, s, [&] { hipLaunchKernelGGL((k_chunk_sweep<1, 32>), dim3(grid1), dim3(256), 0, s, d_slot, d_coef, rec, cz, ntiles, part); });
      }
      printf("record %2d B  chunks %3d (%.2f MB of records each)  tiles %6ld  pieces %7ld (%.1f per camera) : list order %7.1f us   XCD-affine %7.1f us\n",
             rec_bytes, C, (double)Pc * rec_bytes / 1e6, ntiles, pieces, (double)pieces / N, ms0 * 1e3, ms1 * 1e3);
      CHECK(hipFree(d_slot));
      CHECK(hipFree(d_coef));
    }
  }
  return 0;
}
