// pcr_sort.hip -- stable LSD radix sort of (uint64 key, uint32 value) pairs.
// The ONE place a library primitive is used: rocPRIM's device radix sort (plain library sort, like a library
// GEMM); everything else in libpcr_hip.so is hand-written.  Stability matters: points of one voxel keep their
// input order, so the float64 voxel sums are bit-identical to the oracle's (DESIGN.md "voxel").
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include "pcr_internal.h"

// rocPRIM picks a merge sort below 1M items: 9 kernels for 200k keys, but 21 for the 600k keys of the merged voxel pass (two
// kernels per merge level); its Onesweep radix path needs 2 + ceil(bits / 8) = 8 for the ~41 significant bits of those keys.
// Above `onesweep_from` items the Onesweep path is forced (PCR_SORT_ONESWEEP_FROM, default 300000; forcing it for 200k keys was
// measured slower).
using OnesweepAlways = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
static size_t onesweep_from() {
    static const size_t v = getenv("PCR_SORT_ONESWEEP_FROM") ? (size_t)atoll(getenv("PCR_SORT_ONESWEEP_FROM")) : 300000;
    return v;
}

size_t pcr_sort_temp_bytes(size_t n) {
    size_t bytes = 0, bytes2 = 0;
    uint64_t *k = nullptr; uint32_t *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, n, 0u, 64u, (hipStream_t)0, false);
    (void)rocprim::radix_sort_pairs<OnesweepAlways>(nullptr, bytes2, k, k, v, v, n, 0u, 64u, (hipStream_t)0, false);
    return (bytes > bytes2 ? bytes : bytes2) + 256;
}

int pcr_sort_pairs(pcr_context *ctx, void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                   const uint32_t *vals_in, uint32_t *vals_out, size_t n, int end_bit) {
    if (n == 0) return PCR_OK;
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 64) end_bit = 64;
    if (n >= onesweep_from())
        PCR_HIP_CHECK(ctx, rocprim::radix_sort_pairs<OnesweepAlways>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u,
                                                                     (unsigned)end_bit, ctx->stream, false));
    else
        PCR_HIP_CHECK(ctx, rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u,
                                                     (unsigned)end_bit, ctx->stream, false));
    return PCR_OK;
}
