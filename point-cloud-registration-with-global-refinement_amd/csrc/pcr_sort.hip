// pcr_sort.hip -- stable LSD radix sort of (uint64 key, uint32 value) pairs.
// The ONE place a library primitive is used: rocPRIM's device radix sort (plain library sort, like a library
// GEMM); everything else in libpcr_hip.so is hand-written.  Stability matters: points of one voxel keep their
// input order, so the float64 voxel sums are bit-identical to the oracle's (DESIGN.md "voxel").
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "pcr_internal.h"

size_t pcr_sort_temp_bytes(size_t n) {
    size_t bytes = 0;
    uint64_t *k = nullptr; uint32_t *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, n, 0u, 64u, (hipStream_t)0, false);
    return bytes + 256;
}

int pcr_sort_pairs(pcr_context *ctx, void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                   const uint32_t *vals_in, uint32_t *vals_out, size_t n, int end_bit) {
    if (n == 0) return PCR_OK;
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 64) end_bit = 64;
    PCR_HIP_CHECK(ctx, rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u,
                                                 (unsigned)end_bit, ctx->stream, false));
    return PCR_OK;
}
