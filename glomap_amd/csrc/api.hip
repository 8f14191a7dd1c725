// api.hip — context, communicator and misc entry points of the C ABI (include/gsfm.h).
#include <random>

#include "common.hpp"
#include "mt19937.hpp"

using namespace gsfm;

extern "C" int gsfm_version(void) { return GSFM_VERSION; }

extern "C" const char* gsfm_status_string(int status) {
  switch (status) {
    case GSFM_OK: return "ok";
    case GSFM_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GSFM_ERR_HIP: return "HIP runtime error";
    case GSFM_ERR_NO_DEVICE: return "no HIP device";
    case GSFM_ERR_NUMERICAL: return "numerical failure (NaN)";
    case GSFM_ERR_EMPTY_PROBLEM: return "empty problem";
    case GSFM_ERR_NOT_USABLE: return "solution not usable";
    case GSFM_ERR_UNSUPPORTED: return "unsupported configuration";
    case GSFM_ERR_COMM: return "RCCL error";
    default: return "unknown status";
  }
}

extern "C" int gsfm_ctx_create(int device_id, gsfm_ctx** out) {
  if (!out) return GSFM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    fprintf(stderr, "[gsfm] no HIP device available — libgsfm has no CPU fallback\n");
    return GSFM_ERR_NO_DEVICE;
  }
  gsfm_ctx* ctx = new gsfm_ctx();
  int rc = guarded(ctx, nullptr, [&] {
    if (device_id < 0) {
      GSFM_HIP_CHECK(hipGetDevice(&ctx->device));
    } else {
      GSFM_REQUIRE(device_id < count, "device_id out of range");
      ctx->device = device_id;
    }
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    GSFM_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    ctx->num_cus = prop.multiProcessorCount;
    GSFM_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    GSFM_HIP_CHECK(hipEventCreate(&ctx->ev0));
    GSFM_HIP_CHECK(hipEventCreate(&ctx->ev1));
    GSFM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pinned), 4096 * sizeof(double), hipHostMallocDefault));
    if (const char* dd = getenv("GSFM_DUMP_DIR")) ctx->dump_dir = dd;
    return (int)GSFM_OK;
  });
  if (rc != GSFM_OK) {
    gsfm_ctx_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return GSFM_OK;
}

static void peer_close(gsfm_ctx* ctx);

extern "C" void gsfm_ctx_destroy(gsfm_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ra_ws && ctx->ra_ws_free) ctx->ra_ws_free(ctx->ra_ws);
  if (ctx->ra_rig_ws && ctx->ra_rig_ws_free) ctx->ra_rig_ws_free(ctx->ra_rig_ws);
  if (ctx->gp_ws && ctx->gp_ws_free) ctx->gp_ws_free(ctx->gp_ws);
  if (ctx->ba_ws && ctx->ba_ws_free) ctx->ba_ws_free(ctx->ba_ws);
  if (ctx->ba_ws_wide && ctx->ba_ws_wide_free) ctx->ba_ws_wide_free(ctx->ba_ws_wide);
  if (ctx->fl_ws && ctx->fl_ws_free) ctx->fl_ws_free(ctx->fl_ws);
  if (ctx->tr_ws && ctx->tr_ws_free) ctx->tr_ws_free(ctx->tr_ws);
  if (ctx->comm.nccl) (void)ncclCommDestroy(ctx->comm.nccl);
  peer_close(ctx);
  ctx->prof.destroy();
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* gsfm_ctx_last_error(gsfm_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" void* gsfm_ctx_stream(gsfm_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

extern "C" int gsfm_ctx_device_name(gsfm_ctx* ctx, char* buf, size_t buflen) {
  if (!ctx || !buf || buflen == 0) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    hipDeviceProp_t prop;
    GSFM_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_device_alloc(gsfm_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    GSFM_HIP_CHECK(hipMalloc(out, bytes ? bytes : 8));
    return (int)GSFM_OK;
  });
}
extern "C" int gsfm_device_free(gsfm_ctx* ctx, void* ptr) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    if (ptr) GSFM_HIP_CHECK(hipFree(ptr));
    return (int)GSFM_OK;
  });
}
static int copy_sync(gsfm_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (!ctx || (bytes && (!dst || !src))) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    if (bytes) {
      GSFM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
      GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return (int)GSFM_OK;
  });
}
extern "C" int gsfm_memcpy_h2d(gsfm_ctx* ctx, void* d, const void* s, size_t n) { return copy_sync(ctx, d, s, n, hipMemcpyHostToDevice); }
extern "C" int gsfm_memcpy_d2h(gsfm_ctx* ctx, void* d, const void* s, size_t n) { return copy_sync(ctx, d, s, n, hipMemcpyDeviceToHost); }
extern "C" int gsfm_memcpy_d2d(gsfm_ctx* ctx, void* d, const void* s, size_t n) { return copy_sync(ctx, d, s, n, hipMemcpyDeviceToDevice); }
extern "C" int gsfm_ctx_synchronize(gsfm_ctx* ctx) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_ctx_profile_enable(gsfm_ctx* ctx, int enable) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  ctx->prof.enabled = enable != 0;
  return GSFM_OK;
}

extern "C" int gsfm_ctx_lm_trace(gsfm_ctx* ctx, double* out, int32_t max_rows) {
  if (!ctx) return 0;
  const int rows = (int)(ctx->lm_trace.size() / GSFM_LM_TRACE_COLS);
  if (out && max_rows > 0)
    std::memcpy(out, ctx->lm_trace.data(), sizeof(double) * GSFM_LM_TRACE_COLS * (size_t)std::min<int>(rows, max_rows));
  return rows;
}

extern "C" int gsfm_ctx_stats(gsfm_ctx* ctx, int64_t* out, int n, int reset) {
  if (!ctx || (n > 0 && !out)) return GSFM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n && i < GSFM_STAT_COUNT; ++i) out[i] = ctx->stats[i];
  if (reset) std::memset(ctx->stats, 0, sizeof(ctx->stats));
  return GSFM_OK;
}

extern "C" int gsfm_ctx_set_knob(gsfm_ctx* ctx, int knob, int value) {
  if (!ctx || knob < 0 || knob >= GSFM_KNOB_COUNT) return GSFM_ERR_INVALID_ARGUMENT;
  ctx->knob[knob] = value;
  return GSFM_OK;
}

extern "C" int gsfm_ctx_profile_read(gsfm_ctx* ctx, int kernel_id, int64_t* launches, double* total_ms) {
  if (!ctx || kernel_id < 0 || kernel_id >= GSFM_KERNEL_COUNT) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->prof.harvest();
    if (launches) *launches = ctx->prof.launches[kernel_id];
    if (total_ms) *total_ms = ctx->prof.total_ms[kernel_id];
    ctx->prof.launches[kernel_id] = 0;
    ctx->prof.total_ms[kernel_id] = 0.0;
    return (int)GSFM_OK;
  });
}

// ---- RCCL communicator -----------------------------------------------------------------------
static_assert(sizeof(ncclUniqueId) <= GSFM_COMM_ID_BYTES, "ncclUniqueId does not fit GSFM_COMM_ID_BYTES");

extern "C" int gsfm_comm_unique_id(char id[GSFM_COMM_ID_BYTES]) {
  if (!id) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(nullptr, nullptr, [&] {
    ncclUniqueId uid;
    GSFM_NCCL_CHECK(ncclGetUniqueId(&uid));
    std::memset(id, 0, GSFM_COMM_ID_BYTES);
    std::memcpy(id, &uid, sizeof(uid));
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_comm_init(gsfm_ctx* ctx, const char id[GSFM_COMM_ID_BYTES], int rank, int world_size) {
  if (!ctx || !id) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "comm: bad rank/world_size");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->comm.nccl) {
      GSFM_NCCL_CHECK(ncclCommDestroy(ctx->comm.nccl));
      ctx->comm.nccl = nullptr;
    }
    peer_close(ctx);
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    GSFM_NCCL_CHECK(ncclCommInitRank(&ctx->comm.nccl, world_size, uid, rank));
    ctx->comm.rank = rank;
    ctx->comm.world = world_size;
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_selftest_mt19937(uint32_t seed, uint64_t skip, int64_t count, double scale, double* out_fast,
                                     double* out_std) {
  if (count < 0 || (count > 0 && (!out_fast || !out_std))) return GSFM_ERR_INVALID_ARGUMENT;
  gsfm::FastMt19937 fast(seed);
  fast.discard(skip);
  // in two calls with an odd first length: exercises the phase handling of the block path
  const size_t first = count > 3 ? 3 : (size_t)count;
  fast.fill_uniform_pm1(out_fast, first, scale);
  fast.fill_uniform_pm1(out_fast + first, (size_t)count - first, scale);
  std::mt19937 ref;
  ref.seed(seed);
  ref.discard(skip);
  std::uniform_real_distribution<double> uni(-1.0, 1.0);
  for (int64_t i = 0; i < count; ++i) out_std[i] = scale * uni(ref);
  return GSFM_OK;
}

// One RCCL all-reduce of a small device buffer on the ctx stream (works for world_size == 1 too): checks that the
// communicator, the library's stream and its device memory work together in THIS process.
extern "C" int gsfm_comm_selftest(gsfm_ctx* ctx, double* sum_out) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(ctx->comm.nccl != nullptr, "comm selftest: no RCCL communicator attached");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    double* dev = nullptr;
    GSFM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), 256 * sizeof(double)));
    std::vector<double> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1.0 + i;
    GSFM_HIP_CHECK(hipMemcpyAsync(dev, h.data(), 256 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const ncclResult_t r = ncclAllReduce(dev, dev, 256, ncclDouble, ncclSum, ctx->comm.nccl, ctx->stream);
    if (r == ncclSuccess) {
      GSFM_HIP_CHECK(hipMemcpyAsync(h.data(), dev, 256 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    (void)hipFree(dev);
    GSFM_NCCL_CHECK(r);
    bool ok = true;
    for (int i = 0; i < 256; ++i) ok = ok && h[i] == ctx->comm.world * (1.0 + i);
    if (sum_out) *sum_out = h[0];
    if (!ok) throw StatusError(GSFM_ERR_COMM, "comm selftest: wrong all-reduce result");
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_comm_init_host(gsfm_ctx* ctx, gsfm_host_allreduce_fn fn, void* user, int rank, int world_size) {
  if (!ctx || !fn) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "comm: bad rank/world_size");
    if (ctx->comm.nccl) {
      GSFM_NCCL_CHECK(ncclCommDestroy(ctx->comm.nccl));
      ctx->comm.nccl = nullptr;
    }
    peer_close(ctx);
    ctx->comm.host_fn = fn;
    ctx->comm.host_user = user;
    ctx->comm.rank = rank;
    ctx->comm.world = world_size;
    return (int)GSFM_OK;
  });
}

// ---- peer-mailbox transport (peer.hpp) -------------------------------------------------------------
static_assert(sizeof(hipIpcMemHandle_t) <= GSFM_PEER_HANDLE_BYTES, "hipIpcMemHandle_t does not fit GSFM_PEER_HANDLE_BYTES");

static void peer_close(gsfm_ctx* ctx) {
  PeerLink& L = ctx->comm.peer;
  if (!L.open) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (int r = 0; r < L.dev.world; ++r)
    if (r != L.dev.rank && L.connected && L.dev.box[r]) (void)hipIpcCloseMemHandle(L.dev.box[r]);
  if (L.dev.box[L.dev.rank]) (void)hipFree(L.dev.box[L.dev.rank]);
  if (L.counter) (void)hipFree(L.counter);
  if (L.h_err) (void)hipHostFree(L.h_err);
  L = PeerLink{};
}

extern "C" int gsfm_comm_peer_open(gsfm_ctx* ctx, int rank, int world_size, int64_t capacity_doubles,
                                   char handle_out[GSFM_PEER_HANDLE_BYTES]) {
  if (!ctx || !handle_out) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(world_size >= 1 && world_size <= kPeerMaxRanks && rank >= 0 && rank < world_size,
                 "peer comm: rank / world_size out of range (at most 8 ranks)");
    GSFM_REQUIRE(capacity_doubles >= 1024, "peer comm: capacity below 1024 doubles");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    peer_close(ctx);
    PeerLink& L = ctx->comm.peer;
    L.dev.world = world_size;
    L.dev.rank = rank;
    L.dev.cap = ((size_t)capacity_doubles + 511) / 512 * 512;
    for (int r = 0; r < kPeerMaxRanks; ++r) L.dev.box[r] = nullptr;
    const size_t bytes = kPeerHeaderBytes + 2 * (size_t)world_size * L.dev.cap * sizeof(double);
    void* box = nullptr;
    // uncached in L2: the slots are written by the peers over the fabric and read here
    if (hipExtMallocWithFlags(&box, bytes, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      GSFM_HIP_CHECK(hipExtMallocWithFlags(&box, bytes, hipDeviceMallocFinegrained));
    }
    L.dev.box[rank] = static_cast<unsigned char*>(box);
    L.open = true;
    GSFM_HIP_CHECK(hipMemset(box, 0, kPeerHeaderBytes));
    GSFM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&L.counter), sizeof(unsigned)));
    GSFM_HIP_CHECK(hipMemset(L.counter, 0, sizeof(unsigned)));
    GSFM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&L.h_err), sizeof(int), hipHostMallocMapped));
    *L.h_err = 0;
    GSFM_HIP_CHECK(hipDeviceSynchronize());
    const char* to = std::getenv("GSFM_PEER_TIMEOUT_S");
    const double seconds = to && std::atof(to) > 0 ? std::atof(to) : 60.0;
    L.timeout_ticks = (long long)(seconds * 1e8);  // wall_clock64: 100 MHz
    hipIpcMemHandle_t h;
    GSFM_HIP_CHECK(hipIpcGetMemHandle(&h, box));
    std::memset(handle_out, 0, GSFM_PEER_HANDLE_BYTES);
    std::memcpy(handle_out, &h, sizeof(h));
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_comm_peer_connect(gsfm_ctx* ctx, const char* handles) {
  if (!ctx || !handles) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    PeerLink& L = ctx->comm.peer;
    GSFM_REQUIRE(L.open && !L.connected, "peer comm: gsfm_comm_peer_open first (once)");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    L.connected = true;  // (from here on peer_close releases what was mapped)
    for (int r = 0; r < L.dev.world; ++r) {
      if (r == L.dev.rank) continue;
      hipIpcMemHandle_t h;
      std::memcpy(&h, handles + (size_t)r * GSFM_PEER_HANDLE_BYTES, sizeof(h));
      void* p = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        peer_close(ctx);
        throw StatusError(GSFM_ERR_COMM, std::string("peer comm: hipIpcOpenMemHandle failed: ") + hipGetErrorString(e));
      }
      L.dev.box[r] = static_cast<unsigned char*>(p);
    }
    if (ctx->comm.nccl) {
      GSFM_NCCL_CHECK(ncclCommDestroy(ctx->comm.nccl));
      ctx->comm.nccl = nullptr;
    }
    ctx->comm.host_fn = nullptr;
    ctx->comm.rank = L.dev.rank;
    ctx->comm.world = L.dev.world;
    return (int)GSFM_OK;
  });
}

// Checked all-reduce through whatever transport is attached (RCCL, peer mailboxes, host-staged): 256 values, twice in a row
// (both slot sets of the peer transport), sum and max.
static void comm_selftest_any(gsfm_ctx* ctx, double* sum_out) {
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  double* dev = nullptr;
  GSFM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), 256 * sizeof(double)));
  std::vector<double> h(256);
  bool ok = true;
  const int R = ctx->comm.world, me = ctx->comm.rank;
  try {
    for (int round = 0; round < 4; ++round) {
      const int op = round & 1;
      for (int i = 0; i < 256; ++i) h[i] = (1.0 + i) * (me + 1) + round;
      GSFM_HIP_CHECK(hipMemcpyAsync(dev, h.data(), 256 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      allreduce(ctx, dev, 256, op);
      GSFM_HIP_CHECK(hipMemcpyAsync(h.data(), dev, 256 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      for (int i = 0; i < 256; ++i) {
        const double want = op == 0 ? (1.0 + i) * (R * (R + 1) / 2) + (double)round * R : (1.0 + i) * R + round;
        ok = ok && h[i] == want;
      }
    }
  } catch (...) {
    (void)hipFree(dev);
    throw;
  }
  (void)hipFree(dev);
  if (sum_out) *sum_out = (double)R;
  if (ctx->comm.peer.connected && *ctx->comm.peer.h_err) throw StatusError(GSFM_ERR_COMM, "peer all-reduce: a rank did not arrive");
  if (!ok) throw StatusError(GSFM_ERR_COMM, "comm selftest: wrong all-reduce result");
}

extern "C" int gsfm_comm_peer_selftest(gsfm_ctx* ctx, double* world_out) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(ctx->comm.peer.connected, "peer selftest: no peer transport attached");
    comm_selftest_any(ctx, world_out);
    return (int)GSFM_OK;
  });
}

// `repeats` back-to-back all-reduces of n doubles on the ctx stream through the attached transport, timed with HIP events
// around the whole train (after one untimed warm-up): the per-collective cost as a PCG loop sees it.  Collective.
extern "C" int gsfm_comm_allreduce_bench(gsfm_ctx* ctx, int64_t n, int repeats, double* avg_us_out) {
  if (!ctx || n < 1 || repeats < 1 || !avg_us_out) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    double* dev = nullptr;
    GSFM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), (size_t)n * sizeof(double)));
    float ms = 0.f;
    try {
      GSFM_HIP_CHECK(hipMemsetAsync(dev, 0, (size_t)n * sizeof(double), ctx->stream));
      allreduce(ctx, dev, (size_t)n, 0);
      GSFM_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
      for (int i = 0; i < repeats; ++i) allreduce(ctx, dev, (size_t)n, 0);
      GSFM_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
      GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      GSFM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    } catch (...) {
      (void)hipFree(dev);
      throw;
    }
    (void)hipFree(dev);
    if (ctx->comm.peer.connected && *ctx->comm.peer.h_err) throw StatusError(GSFM_ERR_COMM, "peer all-reduce: a rank did not arrive");
    *avg_us_out = 1e3 * ms / repeats;
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_comm_destroy(gsfm_ctx* ctx) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    if (ctx->comm.nccl) {
      GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      GSFM_NCCL_CHECK(ncclCommDestroy(ctx->comm.nccl));
    }
    peer_close(ctx);
    ctx->comm = Comm{};
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_ctx_set_dump_dir(gsfm_ctx* ctx, const char* directory) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  ctx->dump_dir = directory ? directory : "";
  return GSFM_OK;
}
