// Microbenchmark: latency of a hand-written grid barrier (atomic arrive + bounded spin, agent-scope fences)
// across G workgroups of 256 threads on gfx950, with a little neighbour-exchange work between barriers so that
// cross-XCD visibility is actually exercised and verified.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned& target, unsigned nblocks, int* abort_flag) {
  __shared__ int ok_s;
  __syncthreads();
  if (threadIdx.x == 0) {
    target += nblocks;
    int ok = 1;
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22)) { ok = 0; *abort_flag = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

// each round: block b writes buf[b*256+t] = round + b; barrier; reads neighbour block's values and checks
__global__ void __launch_bounds__(256) k_sync(unsigned* counter, int* abort_flag, int* buf, int rounds, int* errors) {
  unsigned target = 0;
  const int nb = gridDim.x;
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    buf[(r & 1) * nb * 256 + blockIdx.x * 256 + threadIdx.x] = r * 1000 + blockIdx.x;
    if (!grid_barrier(counter, target, nb, abort_flag)) return;
    const int other = (blockIdx.x + 1 + (r % 7) * 37) % nb;
    const int v = buf[(r & 1) * nb * 256 + other * 256 + threadIdx.x];
    bad += (v != r * 1000 + other);
  }
  if (bad) atomicAdd(errors, bad);
}

int main() {
  unsigned* counter; int *abort_flag, *buf, *errors;
  CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&abort_flag, 4)); CHECK(hipMalloc(&errors, 4));
  CHECK(hipMalloc(&buf, 2 * 1024 * 256 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int G : {32, 64, 128, 256, 512}) {
    for (int rounds : {1, 1001}) {
      CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(abort_flag, 0, 4)); CHECK(hipMemset(errors, 0, 4));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_sync, dim3(G), dim3(256), 0, 0, counter, abort_flag, buf, rounds, errors);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      int ab, er; CHECK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&er, errors, 4, hipMemcpyDeviceToHost));
      printf("G=%d rounds=%d: %.1f us total, %.2f us per round, abort=%d errors=%d\n", G, rounds, ms * 1e3, ms * 1e3 / rounds, ab, er);
    }
  }
  return 0;
}
