// sr_order.hip -- dequeue order of a batch from the orientation scores.
//
// The alignment kernel's workgroups pull pairs from one queue; a pair is 1/4 of a workgroup's share of C2, so the
// kernel ends with a tail in which fewer and fewer workgroups still work.  Longest-processing-time-first keeps that
// tail short, and once the orientation kernel has run there is a good predictor of a pair's alignment time: the
// orientation score (an edit-distance-like divergence) times the pair's length.  Self pairs (score 0) come last.
// Keys sorted on the device (rocPRIM radix sort, stable: equal costs keep the enumeration order), nothing leaves it.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "sr_internal.h"

__global__ void sr_cost_kernel(const uint32_t *pair_q, const uint32_t *pair_t, const uint32_t *seqlen, const uint8_t *is_reverse,
                               const int32_t *ori_fwd, const int32_t *ori_rev, uint32_t n, uint64_t *keys, uint32_t *vals) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t q = pair_q[i], t = pair_t[i];
        const int32_t s = is_reverse[i] ? ori_rev[i] : ori_fwd[i];
        const uint64_t len = (uint64_t)seqlen[q] + seqlen[t];
        keys[i] = (q == t) ? (uint64_t)seqlen[q] : len * (uint64_t)((s < 0 ? 0 : s) + 16);
        vals[i] = i;
    }
}

extern "C" size_t srk_order_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                         (size_t)n, 0, 64, (hipStream_t)0);
    return bytes;
}

// order[0..n) = indices of the batch's pairs by descending predicted cost
extern "C" int srk_order(const SrAlignArgs *a, uint64_t *keys_in, uint64_t *keys_out, uint32_t *vals_in, void *temp, size_t temp_bytes,
                         uint32_t *order_out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t n = a->npairs;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sr_cost_kernel, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, st,
                       a->pair_q, a->pair_t, a->seqlen, a->is_reverse, a->ori_fwd, a->ori_rev, n, keys_in, vals_in);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_in, keys_out, vals_in, order_out, (size_t)n, 0, 64, st);
    return (int)e;
}
