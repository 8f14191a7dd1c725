// sort.hip — stable device radix sort of (int32 key, int32 value) pairs on the ctx stream.
// Used once per solve to derive the camera-major observation order from the track-major input
// (obsgraph.hpp); rocPRIM is plumbing here, not a hot op.  Kept in its own translation unit so
// the rocPRIM templates are instantiated exactly once.
#include <cstring>

#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.hpp"

namespace gsfm {

// Sorts n pairs by key (stable), keys in [0, 2^key_bits).  tmp is grown on demand.
void sort_pairs_i32(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const int* keys_in, int* keys_out,
                    const int* vals_in, int* vals_out, size_t n, int key_bits) {
  if (n == 0) return;
  size_t bytes = 0;
  const unsigned end_bit = static_cast<unsigned>(key_bits < 1 ? 1 : (key_bits > 32 ? 32 : key_bits));
  GSFM_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u,
                                           end_bit, ctx->stream));
  void* t = tmp.ensure(bytes + 256);
  GSFM_HIP_CHECK(rocprim::radix_sort_pairs(t, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit,
                                           ctx->stream));
}

// Stable DESCENDING sorts (track selection: (length, id) descending = two stable passes, id first).
void sort_pairs_desc_u64(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const unsigned long long* keys_in,
                         unsigned long long* keys_out, const int* vals_in, int* vals_out, size_t n) {
  if (n == 0) return;
  size_t bytes = 0;
  GSFM_HIP_CHECK(rocprim::radix_sort_pairs_desc(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 64u, ctx->stream));
  void* t = tmp.ensure(bytes + 256);
  GSFM_HIP_CHECK(rocprim::radix_sort_pairs_desc(t, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 64u, ctx->stream));
}
void sort_pairs_desc_u32(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const unsigned* keys_in, unsigned* keys_out,
                         const int* vals_in, int* vals_out, size_t n) {
  if (n == 0) return;
  size_t bytes = 0;
  GSFM_HIP_CHECK(rocprim::radix_sort_pairs_desc(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 32u, ctx->stream));
  void* t = tmp.ensure(bytes + 256);
  GSFM_HIP_CHECK(rocprim::radix_sort_pairs_desc(t, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, 32u, ctx->stream));
}

// out[i] = sum of in[0..i) for i in [0, n): pass n = count + 1 with in[count] = 0 to get the total in out[count].
void exclusive_scan_i64(gsfm_ctx* ctx, DevBuf<unsigned char>& tmp, const long* in, long* out, size_t n) {
  if (n == 0) return;
  size_t bytes = 0;
  GSFM_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, 0L, n, rocprim::plus<long>(), ctx->stream));
  void* t = tmp.ensure(bytes + 256);
  GSFM_HIP_CHECK(rocprim::exclusive_scan(t, bytes, in, out, 0L, n, rocprim::plus<long>(), ctx->stream));
}

}  // namespace gsfm
