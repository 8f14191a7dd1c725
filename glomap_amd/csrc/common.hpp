// common.hpp — shared host/device infrastructure of libgsfm (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gsfm.h"
#include "peer.hpp"

namespace gsfm {

// ------------------------------------------------------------------------------------------
// Errors: HIP failures become C++ exceptions inside the library and are turned into
// gsfm_status codes at the C boundary (never thrown across it).
// ------------------------------------------------------------------------------------------
struct HipError : std::runtime_error {
  hipError_t code;
  HipError(hipError_t c, const char* what) : std::runtime_error(what), code(c) {}
};
struct StatusError : std::runtime_error {
  int status;
  StatusError(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

#define GSFM_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      char _buf[512];                                                                     \
      snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
               __FILE__, __LINE__);                                                       \
      throw ::gsfm::HipError(_e, _buf);                                                   \
    }                                                                                     \
  } while (0)

#define GSFM_NCCL_CHECK(expr)                                                              \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess) {                                                               \
      char _buf[512];                                                                      \
      snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), \
               __FILE__, __LINE__);                                                        \
      throw ::gsfm::StatusError(GSFM_ERR_COMM, _buf);                                      \
    }                                                                                      \
  } while (0)

#define GSFM_REQUIRE(cond, msg)                                                  \
  do {                                                                           \
    if (!(cond)) throw ::gsfm::StatusError(GSFM_ERR_INVALID_ARGUMENT, (msg));    \
  } while (0)

// ------------------------------------------------------------------------------------------
// Device buffer (RAII, grows on demand, never shrinks — scratch is owned by the ctx and
// reused across solves so steady-state solves do no hipMalloc).
// ------------------------------------------------------------------------------------------
template <typename T>
class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = nullptr;
    cap_ = 0;
  }
  T* ensure(size_t n) {
    if (n > cap_) {
      release();
      size_t want = n + n / 8 + 64;
      GSFM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ptr_), want * sizeof(T)));
      cap_ = want;
    }
    return ptr_;
  }
  T* get() const { return ptr_; }
  size_t capacity() const { return cap_; }

 private:
  T* ptr_ = nullptr;
  size_t cap_ = 0;
};

inline double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Grid cap for the solver kernels: every solver kernel is launched with <= kMaxBlocks blocks of
// kBlock threads and grid-strides; per-block partial sums live in fixed slots and are re-reduced
// in a fixed order by the consumer kernel => reductions are deterministic for a given grid.
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;
constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------
// ctx
// ------------------------------------------------------------------------------------------
struct Comm {
  ncclComm_t nccl = nullptr;
  int rank = 0;
  int world = 1;
  // host-staged transport (gsfm_comm_init_host): validates the sharded path where RCCL cannot be
  // used (several ranks on ONE device, CI without xGMI).  Never the production path.
  gsfm_host_allreduce_fn host_fn = nullptr;
  void* host_user = nullptr;
  std::vector<double> host_buf;
  // one-shot all-reduce over peer-mapped mailboxes (gsfm_comm_peer_*, peer.hpp): takes every collective once connected
  PeerLink peer;
};

}  // namespace gsfm

namespace gsfm {
// Event-pair pool for per-kernel timing (gsfm_ctx_profile_*).  begin/end record on the ctx stream;
// harvest() must be called after a stream synchronisation and folds the finished pairs into the
// per-kernel totals.
struct KernelProfiler {
  static constexpr int kPool = 2048;
  bool enabled = false;
  std::vector<hipEvent_t> ev;  // 2 * kPool
  std::vector<int> tag;        // kernel id of each used pair
  std::vector<int> iter;       // PCG iteration the pair belongs to (-1: not part of an iteration)
  int used = 0;
  int64_t launches[GSFM_KERNEL_COUNT] = {};
  double total_ms[GSFM_KERNEL_COUNT] = {};
  bool begin(hipStream_t s, int id, int it = -1) {
    if (!enabled || used >= kPool) return false;
    if (ev.empty()) {
      ev.resize(2 * kPool);
      tag.resize(kPool);
      iter.resize(kPool);
      for (auto& e : ev) GSFM_HIP_CHECK(hipEventCreate(&e));
    }
    tag[used] = id;
    iter[used] = it;
    GSFM_HIP_CHECK(hipEventRecord(ev[2 * used], s));
    return true;
  }
  void end(hipStream_t s) {
    GSFM_HIP_CHECK(hipEventRecord(ev[2 * used + 1], s));
    ++used;
  }
  // live_iters: pairs of PCG iterations >= live_iters are dropped — those launches found the solve finished and
  // returned at once (cg.hpp); counting them would dilute the per-launch averages.
  void harvest(int live_iters = 0x7fffffff) {
    for (int i = 0; i < used; ++i) {
      float ms = 0.f;
      if (iter[i] >= live_iters) continue;
      if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) {
        launches[tag[i]]++;
        total_ms[tag[i]] += ms;
      }
    }
    used = 0;
  }
  void destroy() {
    for (auto& e : ev) (void)hipEventDestroy(e);
    ev.clear();
  }
};
}  // namespace gsfm

struct gsfm_ctx {
  int device = 0;
  gsfm::KernelProfiler prof;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  gsfm::Comm comm;
  int num_cus = 256;
  std::string last_error;
  int64_t stats[GSFM_STAT_COUNT] = {};  // which solver paths ran (gsfm_ctx_stats)
  std::vector<double> lm_trace;         // one row of GSFM_LM_TRACE_COLS per LM iteration of the last GP / BA solve (gsfm_ctx_lm_trace)
  int knob[GSFM_KNOB_COUNT] = {};       // diagnostic / A-B knobs (gsfm_ctx_set_knob); 0 = default
  std::string dump_dir;  // non-empty: every solve writes its flat problem + result there (dump.hpp)
  int dump_seq = 0;
  // pinned host staging for small status read-backs
  double* h_pinned = nullptr;  // 4096 doubles
  // opaque per-solver workspaces (allocated lazily, freed by destroy)
  void* ra_ws = nullptr;
  void* gp_ws = nullptr;
  void* ba_ws = nullptr;
  void* ba_ws_wide = nullptr;  // workspace of the 16-wide BA unit (ba_wide.hip)
  void* fl_ws = nullptr;
  void* tr_ws = nullptr;
  void* ra_rig_ws = nullptr;
  void (*ra_rig_ws_free)(void*) = nullptr;
  void (*fl_ws_free)(void*) = nullptr;
  void (*tr_ws_free)(void*) = nullptr;
  void (*ra_ws_free)(void*) = nullptr;
  void (*gp_ws_free)(void*) = nullptr;
  void (*ba_ws_free)(void*) = nullptr;
  void (*ba_ws_wide_free)(void*) = nullptr;
};

namespace gsfm {

// All-reduce (sum) of a device vector over the ranks of the ctx communicator, in place, on the
// ctx stream.  No-op for a single rank.
inline void allreduce(gsfm_ctx* ctx, double* dev, size_t n, int op /* 0 = sum, 1 = max */) {
  if (ctx->comm.world <= 1 || n == 0) return;
  ctx->stats[GSFM_STAT_ALLREDUCES]++;
  if (ctx->comm.peer.connected) {
    if (*ctx->comm.peer.h_err) throw StatusError(GSFM_ERR_COMM, "peer all-reduce: a rank did not arrive within the time limit");
    peer_allreduce(ctx->comm.peer, ctx->stream, dev, n, op);
    return;
  }
  if (ctx->comm.host_fn) {
    std::vector<double>& h = ctx->comm.host_buf;
    h.resize(n);
    GSFM_HIP_CHECK(hipMemcpyAsync(h.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->comm.host_fn(h.data(), (int64_t)n, op, ctx->comm.host_user) != 0)
      throw StatusError(GSFM_ERR_COMM, "host all-reduce callback failed");
    GSFM_HIP_CHECK(hipMemcpyAsync(dev, h.data(), n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return;
  }
  GSFM_NCCL_CHECK(ncclAllReduce(dev, dev, n, ncclDouble, op == 0 ? ncclSum : ncclMax, ctx->comm.nccl, ctx->stream));
}
// After a stream synchronisation that follows collectives: a peer all-reduce that timed out has poisoned its output with
// NaN and raised the host-mapped flag — fail here instead of returning a result computed from it.
inline void comm_check(gsfm_ctx* ctx) {
  if (ctx->comm.world > 1 && ctx->comm.peer.connected && *ctx->comm.peer.h_err)
    throw StatusError(GSFM_ERR_COMM, "peer all-reduce: a rank did not arrive within the time limit");
}
inline void allreduce_sum(gsfm_ctx* ctx, double* dev, size_t n) { allreduce(ctx, dev, n, 0); }
inline void allreduce_max(gsfm_ctx* ctx, double* dev, size_t n) { allreduce(ctx, dev, n, 1); }

// Copies `n` elements host<->device or device<->device depending on the problem's mem space.
template <typename T>
inline void copy_in(gsfm_ctx* ctx, T* dst_dev, const T* src, size_t n, int mem) {
  if (n == 0) return;
  GSFM_HIP_CHECK(hipMemcpyAsync(dst_dev, src, n * sizeof(T),
                                mem == GSFM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                ctx->stream));
}
template <typename T>
inline void copy_out(gsfm_ctx* ctx, T* dst, const T* src_dev, size_t n, int mem) {
  if (n == 0) return;
  GSFM_HIP_CHECK(hipMemcpyAsync(dst, src_dev, n * sizeof(T),
                                mem == GSFM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                ctx->stream));
}
template <typename T>
inline void to_host(gsfm_ctx* ctx, std::vector<T>& dst, const T* src, size_t n, int mem) {
  dst.resize(n);
  if (n == 0) return;
  if (mem == GSFM_MEM_DEVICE) {
    GSFM_HIP_CHECK(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  } else {
    std::memcpy(dst.data(), src, n * sizeof(T));
  }
}

inline int grid_for(size_t work_items, int items_per_block) {
  size_t g = (work_items + items_per_block - 1) / items_per_block;
  if (g < 1) g = 1;
  if (g > static_cast<size_t>(kMaxBlocks)) g = kMaxBlocks;
  return static_cast<int>(g);
}

// Grid for kernels WITHOUT per-block partial slots: one block per `items_per_block` work items,
// capped only by `cap` (the hardware dispatcher streams the blocks; more waves in flight hide the
// dependent-gather latency of the sweeps far better than a grid-stride loop does).
inline int grid_wide(size_t work_items, int items_per_block, size_t cap = (1u << 22)) {
  size_t g = (work_items + items_per_block - 1) / items_per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<int>(g);
}

// Translate exceptions to status codes at the C boundary.
template <typename F>
inline int guarded(gsfm_ctx* ctx, gsfm_report* rep, F&& f) {
  try {
    return f();
  } catch (const HipError& e) {
    if (ctx) ctx->last_error = e.what();
    if (rep) rep->hip_error = static_cast<int32_t>(e.code);
    fprintf(stderr, "[gsfm] %s\n", e.what());
    return GSFM_ERR_HIP;
  } catch (const StatusError& e) {
    if (ctx) ctx->last_error = e.what();
    fprintf(stderr, "[gsfm] %s\n", e.what());
    return e.status;
  } catch (const std::exception& e) {
    if (ctx) ctx->last_error = e.what();
    fprintf(stderr, "[gsfm] %s\n", e.what());
    return GSFM_ERR_INVALID_ARGUMENT;
  }
}

}  // namespace gsfm
