// sr_sketch.hip -- pair sparsification on the device: k-mer bottom-s sketches, all-pairs
// Jaccard, k-nearest / k-farthest selection.
//
// Reference: the grammar `tree:kn,kf,rf[,kmer]` / `auto` / `connectivity:P` / `random:F` is in-tree
// (src/seqrush.rs:356-431) and so are the call sites (AllPairIterator::with_options .. sparsification, :728-735;
// allwave::knn_graph::extract_tree_pairs_separated(&seqs, k_nearest, k_farthest, rand_frac, kmer), :941-947), but
// the implementation (allwave `knn_graph`, mash sketches) is not in the reference tree.  The definition used here is
// this project's own, restated verbatim in oracle/seqrush.c: PARITY UNPINNED.
//
//   k-mer at position i is valid iff its k bytes are in ACGTacgt (case-insensitive codes A=0 C=1 G=2 T=3);
//   canonical = min(forward code, reverse-complement code), first base most significant;
//   hash = splitmix64(canonical ^ k * 0x9E3779B97F4A7C15); hashes equal to 2^64-1 are dropped;
//   sketch = the s = 1000 smallest DISTINCT hashes, ascending;
//   similarity(a, b) = shared / denom over the denom = min(s, |A u B|) smallest elements of the union;
//   x nearer than y  <=>  shared_x * denom_y > shared_y * denom_x, ties: lower index first.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sr_internal.h"

#define SK_WG 1024
#define SK_SENT 0xFFFFFFFFFFFFFFFFULL

__device__ __forceinline__ unsigned long long sk_mix(unsigned long long x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__device__ __forceinline__ int sk_code(uint8_t b) {
    switch (b) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return -1; }
}

// one workgroup per sequence: hash every k-mer into scratch[seq * stride .. +npad), pad with the sentinel
__global__ void __launch_bounds__(SK_WG) sr_kmer_hash_kernel(const uint8_t *bases, const uint64_t *goff, const uint32_t *len,
                                                             int k, unsigned long long *scratch, uint64_t stride, const uint32_t *npad) {
    const uint32_t s = blockIdx.x;
    const uint8_t *b = bases + goff[s];
    const uint32_t L = len[s], N = npad[s];
    unsigned long long *out = scratch + (uint64_t)s * stride;
    const unsigned long long salt = (unsigned long long)k * 0x9E3779B97F4A7C15ULL;
    for (uint32_t i = threadIdx.x; i < N; i += SK_WG) {
        unsigned long long h = SK_SENT;
        if (i + (uint32_t)k <= L) {
            unsigned long long f = 0, r = 0;
            bool ok = true;
            for (int j = 0; j < k; j++) {
                const int c = sk_code(b[i + j]);
                if (c < 0) { ok = false; break; }
                f = (f << 2) | (unsigned long long)c;
                r |= (unsigned long long)(3 - c) << (2 * j);
            }
            if (ok) h = sk_mix((f < r ? f : r) ^ salt);
        }
        out[i] = h;
    }
}

// bitonic sort of scratch[seq] (npad a power of two) in global memory, then the first s distinct values -> sketch
__global__ void __launch_bounds__(SK_WG) sr_sketch_sort_kernel(unsigned long long *scratch, uint64_t stride, const uint32_t *npad,
                                                               int s_max, unsigned long long *sketch, uint32_t *sk_n) {
    __shared__ uint32_t wsum[SK_WG / 64];
    __shared__ uint32_t total;
    const uint32_t seq = blockIdx.x, N = npad[seq];
    unsigned long long *a = scratch + (uint64_t)seq * stride;
    const uint32_t tid = threadIdx.x;
    for (uint32_t size = 2; size <= N; size <<= 1) {
        for (uint32_t st = size >> 1; st > 0; st >>= 1) {
            for (uint32_t t = tid; t < (N >> 1); t += SK_WG) {
                const uint32_t i = ((t / st) * (st << 1)) + (t % st), j = i + st;
                const bool asc = (i & size) == 0;
                const unsigned long long x = a[i], y = a[j];
                if ((x > y) == asc) { a[i] = y; a[j] = x; }
            }
            __syncthreads();
        }
    }
    // first s_max distinct non-sentinel values
    unsigned long long *out = sketch + (uint64_t)seq * (uint64_t)s_max;
    uint32_t count = 0;
    for (uint32_t base = 0; base < N && count < (uint32_t)s_max; base += SK_WG) {
        const uint32_t i = base + tid;
        const unsigned long long v = (i < N) ? a[i] : SK_SENT;
        const bool flag = (i < N) && v != SK_SENT && (i == 0 || a[i - 1] != v);
        uint32_t x = flag ? 1u : 0u;
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        uint32_t pre = 0;
        for (int w = 0; w < wv; w++) pre += wsum[w];
        if (tid == SK_WG - 1) total = pre + x;
        const uint32_t rank = count + pre + x - 1;          // 0-based rank of a flagged element
        if (flag && rank < (uint32_t)s_max) out[rank] = v;
        __syncthreads();
        count += total;
        __syncthreads();
    }
    if (tid == 0) sk_n[seq] = count < (uint32_t)s_max ? count : (uint32_t)s_max;
}

// shared / denom for every i < j (and mirrored): merge walk over the two ascending sketches
__global__ void sr_jaccard_kernel(const unsigned long long *sketch, const uint32_t *sk_n, uint32_t n, int s_max,
                                  uint32_t *shared, uint32_t *denom) {
    const uint64_t total = (uint64_t)n * n;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(p / n), j = (uint32_t)(p % n);
        if (i >= j) continue;
        const unsigned long long *A = sketch + (uint64_t)i * s_max, *B = sketch + (uint64_t)j * s_max;
        const uint32_t na = sk_n[i], nb = sk_n[j];
        uint32_t x = 0, y = 0, sh = 0, dn = 0;
        while (dn < (uint32_t)s_max && (x < na || y < nb)) {
            if (y >= nb || (x < na && A[x] < B[y])) x++;
            else if (x >= na || B[y] < A[x]) y++;
            else { x++; y++; sh++; }
            dn++;
        }
        if (dn == 0) dn = 1;
        shared[(uint64_t)i * n + j] = sh; denom[(uint64_t)i * n + j] = dn;
        shared[(uint64_t)j * n + i] = sh; denom[(uint64_t)j * n + i] = dn;
    }
}

// one thread per row: kn nearest then kf farthest neighbours (each as k selection passes), sel[i][j] = 1
__global__ void sr_knn_select_kernel(const uint32_t *shared, const uint32_t *denom, uint32_t n, int kn, int kf, uint8_t *sel) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *S = shared + (uint64_t)i * n, *D = denom + (uint64_t)i * n;
    uint8_t *row = sel + (uint64_t)i * n;
    for (int pass = 0; pass < kn + kf; pass++) {
        const bool nearest = pass < kn;
        int best = -1;
        for (uint32_t j = 0; j < n; j++) {
            if (j == i || (row[j] & (nearest ? 1 : 2))) continue;
            if (best < 0) { best = (int)j; continue; }
            const unsigned long long l = (unsigned long long)S[j] * D[best], r = (unsigned long long)S[best] * D[j];
            if (nearest ? (l > r) : (l < r)) best = (int)j;
        }
        if (best < 0) break;
        row[best] |= nearest ? 1 : 2;
    }
}

extern "C" int srk_sketch(const uint8_t *bases, const uint64_t *goff, const uint32_t *len, uint32_t n, int k, int s_max,
                          unsigned long long *scratch, uint64_t stride, const uint32_t *npad, unsigned long long *sketch,
                          uint32_t *sk_n, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sr_kmer_hash_kernel, dim3(n), dim3(SK_WG), 0, st, bases, goff, len, k, scratch, stride, npad);
    hipLaunchKernelGGL(sr_sketch_sort_kernel, dim3(n), dim3(SK_WG), 0, st, scratch, stride, npad, s_max, sketch, sk_n);
    return (int)hipGetLastError();
}
extern "C" int srk_jaccard(const unsigned long long *sketch, const uint32_t *sk_n, uint32_t n, int s_max, uint32_t *shared,
                           uint32_t *denom, void *stream) {
    const uint64_t total = (uint64_t)n * n;
    const int nb = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(sr_jaccard_kernel, dim3(nb ? nb : 1), dim3(256), 0, (hipStream_t)stream, sketch, sk_n, n, s_max, shared, denom);
    return (int)hipGetLastError();
}
extern "C" int srk_knn_select(const uint32_t *shared, const uint32_t *denom, uint32_t n, int kn, int kf, uint8_t *sel, void *stream) {
    hipLaunchKernelGGL(sr_knn_select_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, shared, denom, n, kn, kf, sel);
    return (int)hipGetLastError();
}
