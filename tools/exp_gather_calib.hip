// exp_gather_calib.hip — what does a random record gather cost on MI355X, in time and in FETCH_SIZE?
//
// The camera-major halves of the GP / BA operator (k_gp_phaseB, k_ba_phaseB) gather one 64-byte-aligned point record per
// observation.  /opt/skills/guides/MI355X_MICROARCH.md calibrates "FETCH_SIZE x 2" for wide coalesced streams only and says
// to calibrate other access patterns on a known byte count.  This program is that calibration: N random record gathers of
// 32 / 64 / 128 bytes from a 64 MB and a 1 GB array, one lane per record (REC/16 dependent-free 16-byte loads per lane, what
// the sweeps do) or REC/16 lanes per record (one 16-byte load per lane), next to a coalesced 16 B/lane stream of known size.
//
//   hipcc -O3 --offload-arch=gfx950 tools/exp_gather_calib.hip -o tools/exp_gather_calib
//   tools/exp_gather_calib                     -> time per launch, gathers/s, record bytes/s
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o calib -- tools/exp_gather_calib 1
//   python tools/gather_calib_summary.py out/calib_results.db   -> raw FETCH_SIZE bytes per gather, per kernel
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CHECK(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__);         \
      exit(1);                                                                                    \
    }                                                                                             \
  } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {  // splitmix64 finaliser
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

// coalesced reference: every lane reads 16 bytes, consecutive lanes consecutive addresses
template <int TAG>
__global__ void __launch_bounds__(256) k_stream16(const double2* __restrict__ a, long n16, double* __restrict__ out) {
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    const double2 v = a[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;  // (never true: keeps the loads alive without a store stream)
}

// REC bytes per record, LPR lanes per record (1: the lane issues REC/16 loads; REC/16: one load per lane).
// TAG only separates the kernel names of the two array sizes for the profiler.
template <int REC, int LPR, int TAG>
__global__ void __launch_bounds__(256) k_gather(const double2* __restrict__ a, long nrec, long ngather, unsigned long long seed,
                                                double* __restrict__ out) {
  constexpr int Q = REC / 16;      // 16-byte pieces per record
  constexpr int PER = Q / LPR;     // pieces per lane
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long g = t / LPR;          // gather index
  if (g >= ngather) return;
  const int sub = (int)(t % LPR);
  const long r = (long)(mix((unsigned long long)g + seed) % (unsigned long long)nrec);
  const double2* p = a + r * Q + sub * PER;
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const double2 v = p[j];
    acc += v.x + v.y;
  }
  if (LPR == 1) {
    out[g] = acc;  // 8 bytes per gather, coalesced
  } else {
    for (int o = 1; o < LPR; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0) out[g] = acc;
  }
}

// Sorted-ish gathers: the locality a relabelled scene would give — consecutive gathers hit records within a window of
// `window` records (random inside the window).
template <int REC, int TAG>
__global__ void __launch_bounds__(256) k_gather_window(const double2* __restrict__ a, long nrec, long ngather, long window,
                                                       unsigned long long seed, double* __restrict__ out) {
  constexpr int Q = REC / 16;
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngather) return;
  const long base = (long)((double)g / (double)ngather * (double)(nrec - window));
  const long r = base + (long)(mix((unsigned long long)g + seed) % (unsigned long long)window);
  const double2* p = a + r * Q;
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const double2 v = p[j];
    acc += v.x + v.y;
  }
  out[g] = acc;
}

static float time_launches(int reps, hipStream_t s, const std::function<void()>& launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();  // warm
  CHECK(hipStreamSynchronize(s));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return ms / reps;
}

template <int REC, int LPR, int TAG>
static void run_gather(const char* what, const double2* a, size_t bytes, long ngather, double* out, hipStream_t s, int reps) {
  const long nrec = (long)(bytes / REC);
  const long threads = ngather * LPR;
  const int grid = (int)((threads + 255) / 256);
  unsigned long long seed = 0x1234;
  float ms = time_launches(reps, s, [&] {
    hipLaunchKernelGGL((k_gather<REC, LPR, TAG>), dim3(grid), dim3(256), 0, s, a, nrec, ngather, seed, out);
    seed += 0x9e3779b97f4a7c15ull;  // a new random pattern per launch
  });
  printf("%-46s rec %3d B  lanes/rec %d  array %6.0f MB : %8.1f us  %6.2f G gathers/s  %6.2f TB/s of record bytes\n", what, REC, LPR,
         bytes / 1e6, ms * 1e3, ngather / (ms * 1e-3) / 1e9, (double)ngather * REC / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const long ngather = 6'000'000;  // observations of the configs[3] GP problem
  const size_t small = 64ull << 20, big = 1ull << 30;
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  double2* a = nullptr;
  double* out = nullptr;
  CHECK(hipMalloc((void**)&a, big));
  CHECK(hipMalloc((void**)&out, ngather * sizeof(double)));
  CHECK(hipMemset(a, 0, big));
  CHECK(hipMemset(out, 0, ngather * sizeof(double)));
  CHECK(hipDeviceSynchronize());
  {
    const long n16 = (long)(big / 16);
    float ms = time_launches(reps, s, [&] { hipLaunchKernelGGL((k_stream16<1>), dim3(8192), dim3(256), 0, s, a, n16, out); });
    printf("%-46s                              array %6.0f MB : %8.1f us  %6.2f TB/s\n", "k_stream16<1> (coalesced 16 B/lane)", big / 1e6,
           ms * 1e3, big / (ms * 1e-3) / 1e12);
    const long n16s = (long)(small / 16);
    ms = time_launches(reps, s, [&] { hipLaunchKernelGGL((k_stream16<0>), dim3(8192), dim3(256), 0, s, a, n16s, out); });
    printf("%-46s                              array %6.0f MB : %8.1f us  %6.2f TB/s\n", "k_stream16<0> (coalesced 16 B/lane)", small / 1e6,
           ms * 1e3, small / (ms * 1e-3) / 1e12);
  }
  run_gather<32, 1, 0>("k_gather<32,1,0>", a, small, ngather, out, s, reps);
  run_gather<64, 1, 0>("k_gather<64,1,0>", a, small, ngather, out, s, reps);
  run_gather<128, 1, 0>("k_gather<128,1,0>", a, small, ngather, out, s, reps);
  run_gather<64, 4, 0>("k_gather<64,4,0>", a, small, ngather, out, s, reps);
  run_gather<32, 1, 1>("k_gather<32,1,1>", a, big, ngather, out, s, reps);
  run_gather<64, 1, 1>("k_gather<64,1,1>", a, big, ngather, out, s, reps);
  run_gather<128, 1, 1>("k_gather<128,1,1>", a, big, ngather, out, s, reps);
  run_gather<64, 4, 1>("k_gather<64,4,1>", a, big, ngather, out, s, reps);
  run_gather<128, 8, 1>("k_gather<128,8,1>", a, big, ngather, out, s, reps);
  {  // locality: windows of 256 / 4096 records inside the 64 MB array (1 M records of 64 B)
    const long nrec = (long)(small / 64);
    const int grid = (int)((ngather + 255) / 256);
    for (long window : {64L, 1024L, 16384L}) {
      unsigned long long seed = 7;
      float ms = time_launches(reps, s, [&] {
        if (window == 64)
          hipLaunchKernelGGL((k_gather_window<64, 0>), dim3(grid), dim3(256), 0, s, a, nrec, ngather, window, seed, out);
        else if (window == 1024)
          hipLaunchKernelGGL((k_gather_window<64, 1>), dim3(grid), dim3(256), 0, s, a, nrec, ngather, window, seed, out);
        else
          hipLaunchKernelGGL((k_gather_window<64, 2>), dim3(grid), dim3(256), 0, s, a, nrec, ngather, window, seed, out);
        seed += 0x9e3779b97f4a7c15ull;
      });
      printf("k_gather_window<64,%d> window %6ld records                  array %6.0f MB : %8.1f us  %6.2f G gathers/s\n",
             window == 64 ? 0 : window == 1024 ? 1 : 2, window, small / 1e6, ms * 1e3, ngather / (ms * 1e-3) / 1e9);
    }
  }
  CHECK(hipFree(a));
  CHECK(hipFree(out));
  return 0;
}
