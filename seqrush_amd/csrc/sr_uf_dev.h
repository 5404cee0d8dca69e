// sr_uf_dev.h -- device side of the lock-free union-find (uf_rush-0.2.1/src/lib.rs:112-208) and the walk over one pair's
// CIGAR that unites the bases of its match runs (src/seqrush.rs process_alignment / unite_matching_region via
// src/bidirected_union_find.rs:60-98).  Shared by sr_unite_kernel (sr_uf.hip: one pair per workgroup, after the alignment
// kernel) and by the blocked alignment kernel, which since round 4 unites a pair's runs right after it emitted its CIGAR
// (SrAlignArgs::fuse_unite, sr_ctx_run): the partition is the same in any order, and the atomics of one pair hide under the
// other workgroups' tiles instead of running as a latency-bound kernel of their own.
#pragma once
#include <hip/hip_runtime.h>
#include "sr_internal.h"

// ------------------------------------------------------------------ UF
#define UF_PARENT_MASK 0x03FFFFFFFFFFFFFFULL
#define UF_RANK_SHIFT 58

__device__ __forceinline__ unsigned long long uf_load(unsigned long long *nodes, unsigned long long i) {
    return __hip_atomic_load(&nodes[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool uf_cas(unsigned long long *nodes, unsigned long long i,
                                       unsigned long long expect, unsigned long long desired) {
    return __hip_atomic_compare_exchange_strong(&nodes[i], &expect, desired, __ATOMIC_RELAXED,
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// UFRush::find with path halving (uf_rush lib.rs:112-133)
__device__ __forceinline__ unsigned long long uf_find(unsigned long long *nodes, unsigned long long x,
                                                      int &err) {
    unsigned long long x_node = uf_load(nodes, x);
    int guard = 0;
    while (x != (x_node & UF_PARENT_MASK)) {
        const unsigned long long x_parent = x_node & UF_PARENT_MASK;
        const unsigned long long x_parent_node = uf_load(nodes, x_parent);
        const unsigned long long x_pp = x_parent_node & UF_PARENT_MASK;
        const unsigned long long x_new = x_pp | (x_node & ~UF_PARENT_MASK);
        if (x_new != x_node) (void)uf_cas(nodes, x, x_node, x_new);
        x = x_pp;
        x_node = uf_load(nodes, x);
        if (++guard > (1 << 20)) { err |= SR_DEV_ERR_UF_SPIN; break; }
    }
    return x;
}

// UFRush::unite (uf_rush lib.rs:159-208)
__device__ __forceinline__ bool uf_unite(unsigned long long *nodes, unsigned long long x,
                                         unsigned long long y, int &err) {
    for (int guard = 0; guard < (1 << 16); guard++) {
        unsigned long long x_rep = uf_find(nodes, x, err);
        unsigned long long y_rep = uf_find(nodes, y, err);
        if (x_rep == y_rep) return false;
        const unsigned long long x_node = uf_load(nodes, x_rep);
        const unsigned long long y_node = uf_load(nodes, y_rep);
        unsigned long long x_rank = x_node >> UF_RANK_SHIFT, y_rank = y_node >> UF_RANK_SHIFT;
        if (x_rank > y_rank || (x_rank == y_rank && x_rep > y_rep)) {
            unsigned long long tmp = x_rep; x_rep = y_rep; y_rep = tmp;
            tmp = x_rank; x_rank = y_rank; y_rank = tmp;
        }
        const unsigned long long cur = x_rep | (x_rank << UF_RANK_SHIFT);
        const unsigned long long nw = y_rep | (x_rank << UF_RANK_SHIFT);
        if (uf_cas(nodes, x_rep, cur, nw)) {
            if (x_rank == y_rank) {
                const unsigned long long cv = y_rep | (y_rank << UF_RANK_SHIFT);
                const unsigned long long nv = y_rep | ((y_rank + 1) << UF_RANK_SHIFT);
                (void)uf_cas(nodes, y_rep, cv, nv);
            }
            return true;
        }
    }
    err |= SR_DEV_ERR_UF_SPIN;
    return false;
}


// One pair: inclusive scans of (query advance, target advance, united bases) over chunks of NT operations, then one thread
// per united base.  ops / cnt: the pair's CIGAR; q0 / t0: first aligned position; rc: the query was aligned as its
// reverse complement (bidirected_union_find.rs:72-90: position p of the reversed query is base qlen-1-p, other strand).
// All NT threads of the workgroup call it (barriers inside); united / runs are per-thread tallies.
template <int NT>
__device__ __forceinline__ void uf_unite_cigar(const uint32_t *ops, const uint32_t cnt, const unsigned long long qoff,
                                               const unsigned long long toff, const unsigned long long qlen, const bool rc,
                                               const unsigned long long q0, const unsigned long long t0,
                                               const unsigned long long min_match_len, unsigned long long *nodes,
                                               unsigned long long &united, unsigned long long &runs, int &err) {
    __shared__ unsigned sq[NT], st[NT], sm[NT];   // inclusive scans of one chunk
    __shared__ unsigned long long carry_q, carry_t;
    __shared__ unsigned wsum[3][NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { carry_q = q0; carry_t = t0; }
    __syncthreads();
    for (uint32_t base = 0; base < cnt; base += NT) {
        const uint32_t i = base + tid;
        unsigned dq = 0, dt = 0, ml = 0;
        if (i < cnt) {
            const uint32_t op = ops[i] & 15u; const unsigned len = ops[i] >> 4;
            if (op == SR_OP_M) { dq = len; dt = len; if ((unsigned long long)len >= min_match_len) ml = len; }
            else if (op == SR_OP_X) { dq = len; dt = len; }
            else if (op == SR_OP_I) dt = len;       // raw 'I' consumes text (target)
            else dq = len;                           // raw 'D' consumes pattern (query)
        }
        // block inclusive scan of (dq, dt, ml)
        unsigned vq = dq, vt = dt, vm = ml;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned nq = __shfl_up(vq, o, 64), nt = __shfl_up(vt, o, 64), nm = __shfl_up(vm, o, 64);
            if (lane >= o) { vq += nq; vt += nt; vm += nm; }
        }
        if (lane == 63) { wsum[0][wv] = vq; wsum[1][wv] = vt; wsum[2][wv] = vm; }
        __syncthreads();
        unsigned aq = 0, at = 0, am = 0;
        for (int w = 0; w < wv; w++) { aq += wsum[0][w]; at += wsum[1][w]; am += wsum[2][w]; }
        vq += aq; vt += at; vm += am;
        sq[tid] = vq; st[tid] = vt; sm[tid] = vm;
        __syncthreads();
        const unsigned total_m = sm[NT - 1];
        const unsigned long long cq = carry_q, ct = carry_t;
        if (ml) runs++;
        for (unsigned j = tid; j < total_m; j += NT) {
            // op index: first idx with sm[idx] > j
            int lo = 0, hi = NT - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (sm[mid] > j) hi = mid; else lo = mid + 1; }
            const unsigned len = ops[base + lo] >> 4;
            const unsigned within = j - (sm[lo] - len);
            const unsigned long long qpos = cq + (sq[lo] - len) + within;   // query-space index
            const unsigned long long tpos = ct + (st[lo] - len) + within;
            unsigned long long p1, p2 = (toff + tpos) << 1;
            if (rc) p1 = ((qoff + (qlen - 1 - qpos)) << 1) | 1ULL;
            else p1 = (qoff + qpos) << 1;
            if (p1 != p2) uf_unite(nodes, p1, p2, err);
            united++;
        }
        __syncthreads();
        if (tid == 0) { carry_q = cq + sq[NT - 1]; carry_t = ct + st[NT - 1]; }
        __syncthreads();
    }
}
