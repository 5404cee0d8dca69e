// sr_uf.hip -- union-find kernels (unite from CIGAR match runs, init, canonical
// labels, multi-GPU label merge) and the alignment-kernel dispatcher.
// Reference call sites: src/bidirected_union_find.rs:60-98; uf_rush-0.2.1/src/lib.rs:112-208.
#include <hip/hip_runtime.h>
#include <limits.h>
#include "sr_internal.h"
#include "sr_uf_dev.h"
#define WG SR_WG   // unite kernel workgroup size

// SeqRush::new state (seqrush.rs:324-328): N sequential unite(2i, 2i+1) on a
// fresh forest always ends with parent[2i] = 2i+1 (rank 0) and 2i+1 a root
// of rank 1 (uf_rush tie rule: larger index wins).
__global__ void sr_uf_init_kernel(unsigned long long *nodes, unsigned long long total_len,
                                  unsigned long long uf_size) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < uf_size;
         i += stride) {
        unsigned long long v;
        if ((i >> 1) < total_len) v = (i & 1) ? (i | (1ULL << UF_RANK_SHIFT)) : (i + 1);
        else v = i;
        nodes[i] = v;
    }
}

// process_alignment + unite_matching_region, one pair per workgroup
__global__ void __launch_bounds__(WG) sr_unite_kernel(SrUniteArgs a) {
    const int lane = threadIdx.x & 63;
    unsigned long long united = 0, runs = 0;
    int err = 0;
    for (uint32_t pair = blockIdx.x; pair < a.npairs; pair += gridDim.x) {
        if (a.score[pair] < 0 || a.score[pair] > a.max_score[pair]) continue;   // uniform
        const uint32_t q = a.pair_q[pair], t = a.pair_t[pair];
        uf_unite_cigar<WG>(a.cigar_ops + a.cigar_base[pair], a.cigar_cnt[pair], a.seq_goff[q], a.seq_goff[t], a.seqlen[q],
                           a.is_reverse[pair] != 0, a.q_start ? a.q_start[pair] : 0, a.t_start ? a.t_start[pair] : 0,
                           a.min_match_len, a.nodes, united, runs, err);
    }
    // counters
    for (int o = 32; o > 0; o >>= 1) {
        united += __shfl_xor(united, o, 64);
        runs += __shfl_xor(runs, o, 64);
    }
    if (lane == 0) {
        if (united) atomicAdd(&a.counters[4], united);
        if (runs) atomicAdd(&a.counters[5], runs);
    }
    if (err) atomicOr(a.error_flag, err);
}

// canonical labels: minarr[root] = min element, labels[i] = minarr[find(i)]
__global__ void sr_minroot_kernel(unsigned long long *nodes, unsigned long long n,
                                  unsigned long long *minarr, int *error_flag) {
    int err = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long r = uf_find(nodes, i, err);
        atomicMin(&minarr[r], i);
    }
    if (err) atomicOr(error_flag, err);
}
__global__ void sr_fill_kernel(unsigned long long *p, unsigned long long n, unsigned long long v) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void sr_label_kernel(unsigned long long *nodes, unsigned long long n,
                                const unsigned long long *minarr, unsigned long long *labels,
                                int *error_flag) {
    int err = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long r = uf_find(nodes, i, err);
        labels[i] = minarr[r];
    }
    if (err) atomicOr(error_flag, err);
}
// 32-bit label exchange (u32 when 2N+2 < 2^32 -- half the xGMI bytes of the all-gather)
__global__ void sr_label32_kernel(unsigned long long *nodes, unsigned long long n, const unsigned long long *minarr,
                                  uint32_t *labels, int *error_flag) {
    int err = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long r = uf_find(nodes, i, err);
        labels[i] = (uint32_t)minarr[r];
    }
    if (err) atomicOr(error_flag, err);
}
__global__ void sr_merge32_kernel(unsigned long long *nodes, unsigned long long n, const uint32_t *labels, unsigned count,
                                  int *error_flag) {
    int err = 0;
    const unsigned long long total = n * (unsigned long long)count;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const unsigned long long i = j % n;
        const unsigned long long l = labels[j];
        if (l != i && l < n) uf_unite(nodes, i, l, err);
    }
    if (err) atomicOr(error_flag, err);
}
// replay-unite of `count` gathered label arrays
__global__ void sr_merge_kernel(unsigned long long *nodes, unsigned long long n,
                                const unsigned long long *labels, unsigned count, int *error_flag) {
    int err = 0;
    const unsigned long long total = n * (unsigned long long)count;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const unsigned long long i = j % n;
        const unsigned long long l = labels[j];
        if (l != i && l < n) uf_unite(nodes, i, l, err);
    }
    if (err) atomicOr(error_flag, err);
}

// every kernel family is built once per symbol width of the packed sequence buffer (sr_dev_common.h SR_SYMBITS)
#define SRK_DECL_ALIGN(name)                                                                                              \
    extern "C" int name##_s2(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream);     \
    extern "C" int name##_s4(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream);     \
    extern "C" int name##_s8(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream);
SRK_DECL_ALIGN(srk_align_bfs)
SRK_DECL_ALIGN(srk_align_bfs_wide)
SRK_DECL_ALIGN(srk_align_blk)
// dynamic LDS a kernel may ask for: the CU's 160 KB minus the largest static tables of any alignment kernel
extern "C" int srk_align_max_lds(void) { return 160 * 1024 - 30 * 1024; }
extern "C" int srk_orient_s2(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, void *stream);
extern "C" int srk_orient_s4(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, void *stream);
extern "C" int srk_orient_s8(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, void *stream);

extern "C" int srk_align_blkw_s2(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream);
extern "C" int srk_align(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream) {
    if (a->impl == 2 && (nthreads == 64 || (nthreads == 128 && a->kblock == 10 && off16))) {   // one- / two-wave workgroups (2-bit buffers)
        if (a->symbits != 2) return (int)hipErrorInvalidValue;
        return srk_align_blkw_s2(a, nwg, lds_bytes, off16, nthreads, stream);
    }
    if (a->impl == 2) {
        if (a->symbits == 8) return srk_align_blk_s8(a, nwg, lds_bytes, off16, nthreads, stream);
        if (a->symbits == 4) return srk_align_blk_s4(a, nwg, lds_bytes, off16, nthreads, stream);
        return srk_align_blk_s2(a, nwg, lds_bytes, off16, nthreads, stream);
    }
    if (a->impl == 1 && a->ring_scope + 1 > 32) {              // rings deeper than 32 levels: the wide instance
        if (a->symbits == 8) return srk_align_bfs_wide_s8(a, nwg, lds_bytes, off16, nthreads, stream);
        if (a->symbits == 4) return srk_align_bfs_wide_s4(a, nwg, lds_bytes, off16, nthreads, stream);
        return srk_align_bfs_wide_s2(a, nwg, lds_bytes, off16, nthreads, stream);
    }
    if (a->impl == 1) {
        if (a->symbits == 8) return srk_align_bfs_s8(a, nwg, lds_bytes, off16, nthreads, stream);
        if (a->symbits == 4) return srk_align_bfs_s4(a, nwg, lds_bytes, off16, nthreads, stream);
        return srk_align_bfs_s2(a, nwg, lds_bytes, off16, nthreads, stream);
    }
    return (int)hipErrorInvalidValue;
}
extern "C" int srk_orient(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, void *stream) {
    if (a->symbits == 8) return srk_orient_s8(a, nwg, lds_bytes, off16, stream);
    if (a->symbits == 4) return srk_orient_s4(a, nwg, lds_bytes, off16, stream);
    return srk_orient_s2(a, nwg, lds_bytes, off16, stream);
}

extern "C" int srk_unite(const SrUniteArgs *a, int nwg, void *stream) {
    hipLaunchKernelGGL(sr_unite_kernel, dim3(nwg), dim3(WG), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int srk_uf_init(unsigned long long *nodes, uint64_t total_len, uint64_t uf_size, void *stream) {
    const int nb = (int)((uf_size + 255) / 256 > 4096 ? 4096 : (uf_size + 255) / 256);
    hipLaunchKernelGGL(sr_uf_init_kernel, dim3(nb ? nb : 1), dim3(256), 0, (hipStream_t)stream, nodes,
                       (unsigned long long)total_len, (unsigned long long)uf_size);
    return (int)hipGetLastError();
}

extern "C" int srk_labels(unsigned long long *nodes, uint64_t uf_size, unsigned long long *minarr,
                          unsigned long long *labels, int *error_flag, void *stream) {
    const int nb = (int)((uf_size + 255) / 256 > 4096 ? 4096 : (uf_size + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sr_fill_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, minarr,
                       (unsigned long long)uf_size, ~0ULL);
    hipLaunchKernelGGL(sr_minroot_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, nodes,
                       (unsigned long long)uf_size, minarr, error_flag);
    hipLaunchKernelGGL(sr_label_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, nodes,
                       (unsigned long long)uf_size, minarr, labels, error_flag);
    return (int)hipGetLastError();
}

extern "C" int srk_merge(unsigned long long *nodes, uint64_t uf_size, const unsigned long long *labels,
                         uint32_t count, int *error_flag, void *stream) {
    const uint64_t total = uf_size * count;
    const int nb = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(sr_merge_kernel, dim3(nb ? nb : 1), dim3(256), 0, (hipStream_t)stream, nodes,
                       (unsigned long long)uf_size, labels, count, error_flag);
    return (int)hipGetLastError();
}

extern "C" int srk_labels32(unsigned long long *nodes, uint64_t uf_size, unsigned long long *minarr, uint32_t *labels,
                            int *error_flag, void *stream) {
    const int nb = (int)((uf_size + 255) / 256 > 4096 ? 4096 : (uf_size + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sr_fill_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, minarr, (unsigned long long)uf_size, ~0ULL);
    hipLaunchKernelGGL(sr_minroot_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, nodes, (unsigned long long)uf_size, minarr, error_flag);
    hipLaunchKernelGGL(sr_label32_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, nodes, (unsigned long long)uf_size, minarr, labels,
                       error_flag);
    return (int)hipGetLastError();
}

extern "C" int srk_merge32(unsigned long long *nodes, uint64_t uf_size, const uint32_t *labels, uint32_t count, int *error_flag,
                           void *stream) {
    const uint64_t total = uf_size * count;
    const int nb = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(sr_merge32_kernel, dim3(nb ? nb : 1), dim3(256), 0, (hipStream_t)stream, nodes, (unsigned long long)uf_size,
                       labels, count, error_flag);
    return (int)hipGetLastError();
}
