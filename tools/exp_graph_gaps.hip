// exp_graph_gaps.hip — do HIP graphs shorten the gaps between DEPENDENT kernels on this stack (ROCm 7.2, MI355X)?
//
// A PCG iteration of libgsfm is four dependent launches (k_gp_phaseA -> k_gp_phaseB_x -> k_gp_wsum -> k_cg_update); the
// kernel trace shows 4.6 - 9 us of idle time in front of each (profiles/r04_kernel_gaps.txt), ~25 us per iteration.  This
// program times a chain of 4 dependent kernels per "iteration" (each writes what the next one reads; durations set by a
// spin of N clock ticks), 200 iterations, (a) as plain stream launches, (b) captured once into a graph of 800 kernel nodes,
// (c) captured per chunk of 20 iterations incl. the capture + instantiate cost — what cg_solve could do per solve.
//
//   hipcc -O3 --offload-arch=gfx950 tools/exp_graph_gaps.hip -o tools/exp_graph_gaps && tools/exp_graph_gaps
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                            \
    }                                                                                     \
  } while (0)

__global__ void __launch_bounds__(256) k_step(const double* __restrict__ in, double* __restrict__ out, long n, long spin) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long t0 = wall_clock64();
  double v = i < n ? in[i] : 0.0;
  while (wall_clock64() - t0 < spin) v += 1e-30;
  if (i < n) out[i] = v + 1.0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const long n = 1 << 20;  // 8 MB per vector: every kernel dirties 8 MB that the next one reads (cross-XCD traffic as in the solver)
  const int iters = 200;
  double *a = nullptr, *b = nullptr;
  CHECK(hipMalloc((void**)&a, n * sizeof(double)));
  CHECK(hipMalloc((void**)&b, n * sizeof(double)));
  CHECK(hipMemset(a, 0, n * sizeof(double)));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  const int grid = (int)(n / 256);
  for (long spin : {0L, 2000L, 7000L}) {  // wall_clock64 ticks at 100 MHz: 0, 20, 70 us of kernel body
    auto body = [&](int count) {
      for (int it = 0; it < count; ++it) {
        hipLaunchKernelGGL(k_step, dim3(grid), dim3(256), 0, s, a, b, n, spin);
        hipLaunchKernelGGL(k_step, dim3(grid), dim3(256), 0, s, b, a, n, spin);
        hipLaunchKernelGGL(k_step, dim3(grid / 64), dim3(256), 0, s, a, b, n / 64, spin / 8);  // small kernels (k_gp_wsum, k_cg_update)
        hipLaunchKernelGGL(k_step, dim3(grid / 64), dim3(256), 0, s, b, a, n / 64, spin / 8);
      }
    };
    body(10);
    CHECK(hipStreamSynchronize(s));
    double t0 = now();
    body(iters);
    CHECK(hipStreamSynchronize(s));
    const double t_stream = (now() - t0) / iters * 1e6;
    // (b) one graph for all iterations
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    body(iters);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    t0 = now();
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    const double t_graph = (now() - t0) / iters * 1e6;
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    // (c) capture + instantiate + launch per chunk of 20 iterations
    t0 = now();
    for (int c = 0; c < iters / 20; ++c) {
      CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      body(20);
      CHECK(hipStreamEndCapture(s, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CHECK(hipGraphLaunch(ge, s));
      CHECK(hipStreamSynchronize(s));
      CHECK(hipGraphExecDestroy(ge));
      CHECK(hipGraphDestroy(g));
    }
    const double t_chunk = (now() - t0) / iters * 1e6;
    printf("kernel body %5.1f / %4.1f us: per iteration of 4 dependent kernels — stream launches %7.1f us | one graph %7.1f us | "
           "graph per 20 iterations incl. capture + instantiate %7.1f us\n", spin / 100.0, spin / 800.0, t_stream, t_graph, t_chunk);
  }
  return 0;
}
