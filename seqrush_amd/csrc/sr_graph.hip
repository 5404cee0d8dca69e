// sr_graph.hip -- graph induction from the union-find on the device.
//
// O(N) formulation of build_bidirected_graph_with_options (src/bidirected_builder.rs:17-289) for a
// quiescent union-find given as canonical labels (labels[p] = smallest Pos of p's component):
//   * SeqRush::new unites the two strands of every base (src/seqrush.rs:324-328), so both Pos of a base
//     carry the same label: a node is a component, its key is the label;
//   * node ids follow first-encounter order over sequences / positions (:29-41, :154-157):
//     first[key] = min position (atomicMin), id = 1 + exclusive scan of the "first encounter" flags;
//   * node base = base at offset(label) (:176-182); a step is reversed iff node base and sequence base
//     are complementary (:190-203);
//   * edges: consecutive steps of a path, deduplicated against themselves and their reverse complement,
//     first orientation and first-seen order kept (src/bidirected_ops.rs:813-825): open-addressing hash
//     table keyed by min((from,to), (to^1,from^1)) holding the smallest position, then flag + scan + emit.
// The host (sr_ctx_build_gfa) formats the GFA text from the compact arrays; the text is byte-identical to
// sr_build_gfa() on the downloaded labels (tests/test_gpu_parity.py::test_graph_induction_on_device).
#include <hip/hip_runtime.h>
#include <limits.h>
#include "sr_internal.h"

#define GI_BLOCK 256
#define GI_ITEMS 4
#define GI_TILE (GI_BLOCK * GI_ITEMS)

__device__ __forceinline__ unsigned gi_upper(unsigned b) { return (b >= 'a' && b <= 'z') ? b - 32u : b; }
__device__ __forceinline__ bool gi_complementary(unsigned a, unsigned b) {
    a = gi_upper(a); b = gi_upper(b);
    return (a == 'A' && b == 'T') || (a == 'T' && b == 'A') || (a == 'C' && b == 'G') || (a == 'G' && b == 'C');
}

__global__ void gi_fill_u64(unsigned long long *p, unsigned long long n, unsigned long long v) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void gi_fill_u32(uint32_t *p, unsigned long long n, uint32_t v) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// first[label] = smallest position carrying it
__global__ void gi_first(const unsigned long long *labels, unsigned long long N, unsigned long long ufn,
                         unsigned long long *first, int *error_flag) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += stride) {
        const unsigned long long lf = labels[g << 1], lr = labels[(g << 1) | 1];
        if (lf != lr || lf >= ufn || (lf >> 1) >= N) { atomicOr(error_flag, SR_DEV_ERR_GRAPH); continue; }
        atomicMin(&first[lf], g);
    }
}
__global__ void gi_node_flags(const unsigned long long *labels, unsigned long long N, const unsigned long long *first,
                              uint32_t *flag) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += stride)
        flag[g] = (first[labels[g << 1]] == g) ? 1u : 0u;
}
// steps[g] = (node id << 1) | reversed; node_base[id - 1]
__global__ void gi_steps(const unsigned long long *labels, const uint8_t *bases, unsigned long long N,
                         const unsigned long long *first, const uint32_t *nid, uint32_t *steps, uint8_t *node_base) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += stride) {
        const unsigned long long key = labels[g << 1];
        const unsigned long long g0 = first[key];
        const uint32_t id0 = nid[g0];                        // id - 1
        const unsigned nb = bases[key >> 1];
        steps[g] = ((id0 + 1u) << 1) | (gi_complementary(nb, bases[g]) ? 1u : 0u);
        if (g0 == g) node_base[id0] = (uint8_t)nb;
    }
}

__device__ __forceinline__ unsigned long long gi_mix(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
// hash insert of every path edge; val[slot] = smallest position of an edge with that canonical key
__global__ void gi_edge_insert(const uint32_t *steps, const uint8_t *islast, unsigned long long N,
                               unsigned long long *keys, uint32_t *vals, unsigned long long mask, uint32_t *eslot,
                               int *error_flag) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += stride) {
        if (islast[g]) { eslot[g] = 0xffffffffu; continue; }
        const unsigned long long from = steps[g], to = steps[g + 1];
        const unsigned long long k1 = (from << 32) | to, k2 = ((to ^ 1ull) << 32) | (from ^ 1ull);
        const unsigned long long key = k1 < k2 ? k1 : k2;
        unsigned long long h = gi_mix(key) & mask;
        bool done = false;
        for (unsigned long long probe = 0; probe <= mask; probe++) {
            const unsigned long long prev = atomicCAS(&keys[h], ~0ull, key);
            if (prev == ~0ull || prev == key) { atomicMin(&vals[h], (uint32_t)g); eslot[g] = (uint32_t)h; done = true; break; }
            h = (h + 1) & mask;
        }
        if (!done) { atomicOr(error_flag, SR_DEV_ERR_GRAPH); eslot[g] = 0xffffffffu; }
    }
}
__global__ void gi_edge_flags(const uint32_t *eslot, const uint32_t *vals, unsigned long long N, uint32_t *flag) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += stride) {
        const uint32_t s = eslot[g];
        flag[g] = (s != 0xffffffffu && vals[s] == (uint32_t)g) ? 1u : 0u;
    }
}
__global__ void gi_edge_emit(const uint32_t *steps, const uint32_t *flag, const uint32_t *epos, unsigned long long N,
                             unsigned long long *edges) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += stride)
        if (flag[g]) edges[epos[g]] = ((unsigned long long)steps[g] << 32) | (unsigned long long)steps[g + 1];
}

// ---- exclusive scan of uint32 flags: tile sums -> scan of the tile sums (one block) -> scan with offsets
__device__ __forceinline__ uint32_t gi_block_scan(uint32_t v, uint32_t *total) {   // inclusive over the block
    __shared__ uint32_t wsum[GI_BLOCK / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t n = __shfl_up(v, o, 64); if (lane >= o) v += n; }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    uint32_t add = 0, tot = 0;
    for (int w = 0; w < GI_BLOCK / 64; w++) { if (w < wv) add += wsum[w]; tot += wsum[w]; }
    __syncthreads();
    *total = tot;
    return v + add;
}
__global__ void __launch_bounds__(GI_BLOCK) gi_scan_tiles(const uint32_t *in, unsigned long long n, uint32_t *tile_sum) {
    const unsigned long long base = (unsigned long long)blockIdx.x * GI_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < GI_ITEMS; i++) {
        const unsigned long long idx = base + (unsigned long long)threadIdx.x * GI_ITEMS + i;
        if (idx < n) s += in[idx];
    }
    uint32_t tot;
    (void)gi_block_scan(s, &tot);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(GI_BLOCK) gi_scan_sums(uint32_t *tile_sum, unsigned long long ntiles, uint32_t *grand) {
    uint32_t carry = 0;
    for (unsigned long long base = 0; base < ntiles; base += GI_BLOCK) {
        const unsigned long long idx = base + threadIdx.x;
        const uint32_t v = idx < ntiles ? tile_sum[idx] : 0u;
        uint32_t tot;
        const uint32_t inc = gi_block_scan(v, &tot);
        if (idx < ntiles) tile_sum[idx] = carry + inc - v;     // exclusive
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand = carry;
}
__global__ void __launch_bounds__(GI_BLOCK) gi_scan_apply(const uint32_t *in, unsigned long long n, const uint32_t *tile_sum,
                                                         uint32_t *out) {
    const unsigned long long base = (unsigned long long)blockIdx.x * GI_TILE + (unsigned long long)threadIdx.x * GI_ITEMS;
    uint32_t v[GI_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < GI_ITEMS; i++) { v[i] = (base + i < n) ? in[base + i] : 0u; s += v[i]; }
    uint32_t tot;
    const uint32_t inc = gi_block_scan(s, &tot);
    uint32_t run = tile_sum[blockIdx.x] + inc - s;
#pragma unroll
    for (int i = 0; i < GI_ITEMS; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
}

static int gi_grid(unsigned long long n) {
    unsigned long long b = (n + GI_BLOCK - 1) / GI_BLOCK;
    return (int)(b > 8192 ? 8192 : (b ? b : 1));
}
static void gi_exclusive_scan(const uint32_t *in, unsigned long long n, uint32_t *out, uint32_t *tile_sum, uint32_t *grand,
                              hipStream_t st) {
    const unsigned long long ntiles = (n + GI_TILE - 1) / GI_TILE;
    hipLaunchKernelGGL(gi_scan_tiles, dim3((unsigned)ntiles), dim3(GI_BLOCK), 0, st, in, n, tile_sum);
    hipLaunchKernelGGL(gi_scan_sums, dim3(1), dim3(GI_BLOCK), 0, st, tile_sum, ntiles, grand);
    hipLaunchKernelGGL(gi_scan_apply, dim3((unsigned)ntiles), dim3(GI_BLOCK), 0, st, in, n, tile_sum, out);
}

// All device pointers; `first` has uf_size entries, hash table keys/vals have hcap (power of two) entries,
// flag/nid/eslot/steps N entries, tile_sum ceil(N/1024)+1, counts[2] = {nodes, edges}.
extern "C" int srk_graph_induce(const unsigned long long *labels, const uint8_t *bases, const uint8_t *islast,
                                uint64_t N, uint64_t uf_size, unsigned long long *first, uint32_t *flag, uint32_t *nid,
                                uint32_t *steps, uint8_t *node_base, unsigned long long *hkeys, uint32_t *hvals,
                                uint64_t hcap, uint32_t *eslot, unsigned long long *edges, uint32_t *tile_sum,
                                uint32_t *counts, int *error_flag, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N == 0 || N >= 0x7fffffffULL) return -1;
    hipLaunchKernelGGL(gi_fill_u64, dim3(gi_grid(uf_size)), dim3(GI_BLOCK), 0, st, first, (unsigned long long)uf_size, ~0ull);
    hipLaunchKernelGGL(gi_first, dim3(gi_grid(N)), dim3(GI_BLOCK), 0, st, labels, (unsigned long long)N,
                       (unsigned long long)uf_size, first, error_flag);
    hipLaunchKernelGGL(gi_node_flags, dim3(gi_grid(N)), dim3(GI_BLOCK), 0, st, labels, (unsigned long long)N, first, flag);
    gi_exclusive_scan(flag, N, nid, tile_sum, counts + 0, st);
    hipLaunchKernelGGL(gi_steps, dim3(gi_grid(N)), dim3(GI_BLOCK), 0, st, labels, bases, (unsigned long long)N, first, nid,
                       steps, node_base);
    hipLaunchKernelGGL(gi_fill_u64, dim3(gi_grid(hcap)), dim3(GI_BLOCK), 0, st, hkeys, (unsigned long long)hcap, ~0ull);
    hipLaunchKernelGGL(gi_fill_u32, dim3(gi_grid(hcap)), dim3(GI_BLOCK), 0, st, hvals, (unsigned long long)hcap, 0xffffffffu);
    hipLaunchKernelGGL(gi_edge_insert, dim3(gi_grid(N)), dim3(GI_BLOCK), 0, st, steps, islast, (unsigned long long)N, hkeys,
                       hvals, (unsigned long long)(hcap - 1), eslot, error_flag);
    hipLaunchKernelGGL(gi_edge_flags, dim3(gi_grid(N)), dim3(GI_BLOCK), 0, st, eslot, hvals, (unsigned long long)N, flag);
    gi_exclusive_scan(flag, N, nid, tile_sum, counts + 1, st);       // nid reused as edge positions
    hipLaunchKernelGGL(gi_edge_emit, dim3(gi_grid(N)), dim3(GI_BLOCK), 0, st, steps, flag, nid, (unsigned long long)N, edges);
    return (int)hipGetLastError();
}
