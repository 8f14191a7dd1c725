// peer.hpp — one-shot all-reduce over peer-mapped mailboxes (gsfm_comm_peer_*, include/gsfm.h).
//
// The vectors the solvers all-reduce are small (3 N, 6 N + 8 K doubles: 0.2 - 1.2 MB at configs[3]) and there is one per
// PCG iteration, so the collective is latency bound: a ring moves 2 (R-1) hops and RCCL's floor on such sizes is tens of
// microseconds, as long as a whole sweep at 8 ranks.  xGMI is a full mesh of point-to-point links, which
// suits the one-shot scheme: every rank owns a MAILBOX in its HBM with one slot per rank, all ranks map all mailboxes
// (hipIpcOpenMemHandle: peer access over xGMI between GPUs, plain device memory between processes on one GPU), and an
// all-reduce is
//   push   every rank writes its vector into ITS slot of EVERY mailbox (R-1 concurrent link transfers of n doubles each),
//          fence, then the last workgroup to finish raises the rank's flag in every mailbox to the sequence number;
//   sum    every rank waits until all R flags of its own mailbox carry the sequence number and adds the R slots IN RANK
//          ORDER — the same order everywhere, so the replicated result is bit-identical on every rank, which the solvers
//          rely on (all ranks take the same convergence / step decisions).
// Two slot sets alternate with the parity of the sequence number: a rank can only start all-reduce s+2 after its own
// all-reduce s+1 completed, which needed every peer's s+1 flag, which every peer raises after its sum of s (stream order) —
// so nobody is still reading set (s & 1) when it is written again.  No barrier, no host involvement, two launches.
// A rank that never arrives is a bounded spin (GSFM_PEER_TIMEOUT_S, default 60 s), then an error flag in host-mapped
// memory that the next collective turns into GSFM_ERR_COMM.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <vector>

namespace gsfm {

constexpr int kPeerMaxRanks = 8;
constexpr size_t kPeerHeaderBytes = 4096;  // flags: [2 parities][kPeerMaxRanks] u64, padded

struct PeerDev {
  int world = 0, rank = 0;
  size_t cap = 0;                      // doubles per slot
  unsigned char* box[kPeerMaxRanks];   // mailbox of rank r as mapped into this process (box[rank] = the local one)
};
__device__ __forceinline__ unsigned long long* peer_flags(unsigned char* box, int parity, int world) {
  return reinterpret_cast<unsigned long long*>(box) + (size_t)parity * kPeerMaxRanks;
}
__device__ __forceinline__ double* peer_slot(unsigned char* box, int parity, int r, int world, size_t cap) {
  return reinterpret_cast<double*>(box + kPeerHeaderBytes) + ((size_t)parity * world + r) * cap;
}

struct PeerLink {
  PeerDev dev{};
  bool open = false, connected = false;
  unsigned long long seq = 0;
  unsigned* counter = nullptr;   // device: workgroups of the running push that have finished
  int* h_err = nullptr;          // host-mapped: set by a sum that timed out
  long long timeout_ticks = 0;   // of the 100 MHz wall clock
};

// x (n doubles) into slot `rank` of every mailbox; the last workgroup raises the flags.
static __global__ void __launch_bounds__(256) k_peer_push(PeerDev pd, const double* __restrict__ x, size_t n, unsigned long long seq,
                                                   unsigned* __restrict__ counter) {
  const int parity = (int)(seq & 1);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double v = x[i];
    for (int r = 0; r < pd.world; ++r) peer_slot(pd.box[r], parity, pd.rank, pd.world, pd.cap)[i] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      *counter = 0;
      __threadfence_system();
      for (int r = 0; r < pd.world; ++r)
        __hip_atomic_store(peer_flags(pd.box[r], parity, pd.world) + pd.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// x = reduction over the ranks' slots of the local mailbox, in rank order (op 0 = sum, 1 = max)
static __global__ void __launch_bounds__(256) k_peer_sum(PeerDev pd, double* __restrict__ x, size_t n, unsigned long long seq, int op,
                                                  long long timeout_ticks, int* __restrict__ err) {
  __shared__ int late;
  const int parity = (int)(seq & 1);
  unsigned char* mine = pd.box[pd.rank];
  if (threadIdx.x == 0) late = 0;
  __syncthreads();
  if ((int)threadIdx.x < pd.world) {
    const unsigned long long* f = peer_flags(mine, parity, pd.world) + threadIdx.x;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > timeout_ticks) {
        late = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if (late) {  // a rank never arrived: the result must not look like one (NaN poisons every later decision), the host is told
    if (threadIdx.x == 0) *err = 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = __builtin_nan("");
    return;
  }
  // the threads that read the slots are not the ones that acquired the flags: every one of them orders its loads behind the
  // peers' stores itself (system scope; outside the formal model otherwise once the mailboxes are L2-cached allocations)
  __threadfence_system();
  const double* s0 = peer_slot(mine, parity, 0, pd.world, pd.cap);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double acc = __builtin_nontemporal_load(s0 + i);
    for (int r = 1; r < pd.world; ++r) {
      const double v = __builtin_nontemporal_load(s0 + (size_t)r * pd.cap + i);
      acc = op == 0 ? acc + v : (v > acc ? v : acc);
    }
    x[i] = acc;
  }
}

inline int peer_grid(size_t n) {
  const size_t g = (n + 1023) / 1024;
  return (int)(g < 1 ? 1 : (g > 128 ? 128 : g));
}

// in place, on stream s; vectors longer than a slot go in pieces
inline void peer_allreduce(PeerLink& L, hipStream_t s, double* dev, size_t n, int op) {
  for (size_t o = 0; o < n; o += L.dev.cap) {
    const size_t m = n - o < L.dev.cap ? n - o : L.dev.cap;
    ++L.seq;
    hipLaunchKernelGGL(k_peer_push, dim3(peer_grid(m)), dim3(256), 0, s, L.dev, (const double*)(dev + o), m, L.seq, L.counter);
    hipLaunchKernelGGL(k_peer_sum, dim3(peer_grid(m)), dim3(256), 0, s, L.dev, dev + o, m, L.seq, op, L.timeout_ticks, L.h_err);
  }
}

}  // namespace gsfm
