// sr_dev_common.h -- device helpers shared by the alignment kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include "sr_internal.h"

#define NULLV SR_NULL_OFF
// Symbol width of the packed sequence buffer: 2 bits (inputs over upper-case ACGT), 4 bits (<= 16 distinct
// bytes, e.g. ACGTN + soft-masked lower case) or 8 bits (raw bytes).  The reference compares raw bytes
// (src/seqrush.rs:1162-1176, 1268-1283), so the host maps bytes to codes injectively and every kernel
// translation unit is compiled once per width (-DSR_SYMBITS=N, namespace sr_sN).
#ifndef SR_SYMBITS
#define SR_SYMBITS 2
#endif
#if SR_SYMBITS == 2
#define SR_SYM_LOG 1
#define SR_WIN 16
#define SR_WIN_LOG 4
#define SR_NS sr_s2
#define SRK_NAME(x) x##_s2
#elif SR_SYMBITS == 4
#define SR_SYM_LOG 2
#define SR_WIN 8
#define SR_WIN_LOG 3
#define SR_NS sr_s4
#define SRK_NAME(x) x##_s4
#elif SR_SYMBITS == 8
#define SR_SYM_LOG 3
#define SR_WIN 4
#define SR_WIN_LOG 2
#define SR_NS sr_s8
#define SRK_NAME(x) x##_s8
#else
#error "SR_SYMBITS must be 2, 4 or 8"
#endif
// words of one packed sequence copy incl. its two pad words
#define SR_SEQ_WORDS(len) ((((len) + SR_WIN - 1) >> SR_WIN_LOG) + 2)
#ifndef SR_MIN_WAVES
#define SR_MIN_WAVES 4
#endif

extern __shared__ uint32_t lds_seq[];     // 3 regions of max_words: P fwd, P rc, T

#define RFL(x) __builtin_amdgcn_readfirstlane(x)
__device__ __forceinline__ unsigned long long rfl64(unsigned long long v) {
    const unsigned lo = RFL((unsigned)v), hi = RFL((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Explicit address spaces: pointers that travel inside a by-value kernel
// argument struct are generic ("flat") for hipcc; flat loads cost a VGPR pair
// per address and the slow path.  GP = global (HBM) pointer, LP = LDS pointer.
template <typename T> using GP = T __attribute__((address_space(1))) *;
typedef const uint32_t __attribute__((address_space(3))) *LP;

__device__ __forceinline__ int reach(const SrPen &p, int s, int begin) {
    int r;
    if (begin == SR_C_M) {
        r = (s >= p.o1 + p.e1) ? (s - p.o1) / p.e1 : 0;
        if (p.two && s >= p.o2 + p.e2) r = max(r, (s - p.o2) / p.e2);
    } else {
        r = s / p.e1;
        if (p.two) r = max(r, s / p.e2);
    }
    return r;
}

__device__ __forceinline__ int bnd(int c, unsigned lim) {
    return ((unsigned)c > lim) ? NULLV : c;
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// SR_WIN symbols starting at symbol i (SR_SYMBITS bits each, symbol i in the low bits)
__device__ __forceinline__ uint32_t win_fwd(LP w, int i) {
    const int wi = i >> SR_WIN_LOG, sh = (i & (SR_WIN - 1)) << SR_SYM_LOG;
    const uint64_t v = ((uint64_t)w[wi + 1] << 32) | (uint64_t)w[wi];
    return (uint32_t)(v >> sh);
}
// SR_WIN symbols ending at symbol i (symbol i in the high bits)
__device__ __forceinline__ uint32_t win_rev(LP w, int i) {
    return win_fwd(w, i - (SR_WIN - 1));
}

// number of equal bases walking forward from (pi, ti), at most n
__device__ __forceinline__ int ext_fwd(LP P, LP T, int pi, int ti, int n) {
    int tot = 0;
    while (tot < n) {
        const uint32_t x = win_fwd(P, pi + tot) ^ win_fwd(T, ti + tot);
        int c = x ? ((__ffs((int)x) - 1) >> SR_SYM_LOG) : SR_WIN;
        c = min(c, n - tot);
        tot += c;
        if (x) break;
    }
    return tot;
}
// same walking backward from (pi, ti) inclusive
__device__ __forceinline__ int ext_rev(LP P, LP T, int pi, int ti, int n) {
    int tot = 0;
    while (tot < n) {
        const uint32_t x = win_rev(P, pi - tot) ^ win_rev(T, ti - tot);
        int c = x ? (__clz((int)x) >> SR_SYM_LOG) : SR_WIN;
        c = min(c, n - tot);
        tot += c;
        if (x) break;
    }
    return tot;
}

// Four adjacent diagonals per thread: the 2-byte cells of one ring row are
// moved with 8-byte (int16) / 16-byte (int32) accesses -- the vector memory
// pipe is paid per wave instruction, not per byte.
template <typename OT> using V4 = OT __attribute__((ext_vector_type(4)));

template <typename OT>
struct GroupIn {
    V4<OT> mo1, i1, d1, mo2, i2, d2, mx;
    int mo1L, mo1R, i1L, d1R, mo2L, mo2R, i2L, d2R;
};

// ---- batched window reads of the blocked tiles (sr_align_blk.inc, sr_orient.hip) ----------------------------------
#ifndef SR_LDS_WRAP
#define SR_LDS_WRAP 0x1fffcu
#endif
#ifndef SR_NULL_NOEXT
#define SR_NULL_NOEXT 0           // diagnostic builds (scripts/lds_oob): a cell below 0 gains nothing from its windows
#endif
// SR_WIN symbols starting at LDS-wide symbol index S (symbol 0 = the low bits of the word at LDS address 0)
__device__ __forceinline__ uint32_t win_sym(int S) {
    typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) W2;
    // (the mask aligns the address to a word and wraps it into the first 128 KB: a cell that does not extend may point
    // anywhere.  Reads beyond the workgroup's allocation return 0 and disturb nothing -- scripts/lds_oob)
    const uint32_t addr = ((uint32_t)S >> (SR_WIN_LOG - 2)) & SR_LDS_WRAP;
    const W2 w = *(const W2 __attribute__((address_space(3))) *)(uintptr_t)addr;
    return __builtin_amdgcn_alignbit(w.y, w.x, (uint32_t)S << SR_SYM_LOG);
}
// the two words win_sym() takes its window from
__device__ __forceinline__ void win_words(int S, uint32_t &lo, uint32_t &hi) {
    typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) W2;
    // (the mask aligns the address to a word and wraps it into the first 128 KB: a cell that does not extend may point
    // anywhere.  Reads beyond the workgroup's allocation return 0 and disturb nothing -- scripts/lds_oob)
    const uint32_t addr = ((uint32_t)S >> (SR_WIN_LOG - 2)) & SR_LDS_WRAP;
    const W2 w = *(const W2 __attribute__((address_space(3))) *)(uintptr_t)addr;
    lo = w.x; hi = w.y;
}
// the same for an LDS-wide BIT index (symbol index << SR_SYM_LOG): the index is also v_alignbit's shift amount
__device__ __forceinline__ void win_words_bit(int Bx, uint32_t &lo, uint32_t &hi) {
    typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) W2;
    const uint32_t addr = ((uint32_t)Bx >> 3) & SR_LDS_WRAP;
    const W2 w = *(const W2 __attribute__((address_space(3))) *)(uintptr_t)addr;
    lo = w.x; hi = w.y;
}
// sext(16-bit half HI of w) * 2^SR_SYM_LOG + c in one instruction (v_mad_i32_i16): bit index of a packed cell's window
template <int HI> __device__ __forceinline__ int bit_index(uint32_t w, int c) {
    int r;
    if (HI) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "n"(1 << SR_SYM_LOG), "v"(c));
    else asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(w), "n"(1 << SR_SYM_LOG), "v"(c));
    return r;
}
template <int HI> __device__ __forceinline__ int bit_index_u(uint32_t w, int c) {      // c wave-uniform (SGPR)
    int r;
    if (HI) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "n"(1 << SR_SYM_LOG), "s"(c));
    else asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(w), "n"(1 << SR_SYM_LOG), "s"(c));
    return r;
}
// equal leading symbols of an XOR-ed window: 2^31-1 >> .. (i.e. "all") when it is zero
// (v_ffbl_b32 returns -1 for 0; __ffs() - 1 computes the same through a compare and a select the hardware does not need)
__device__ __forceinline__ unsigned ffs_sym(uint32_t xw) {
    unsigned r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(xw)); return r >> SR_SYM_LOG;
}
__device__ __forceinline__ unsigned min3u(unsigned a, unsigned b, unsigned c) {
    unsigned r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
// v in the lanes of the mask, 0 elsewhere (the mask is a ballot: no per-lane flag register crosses the loop)
__device__ __forceinline__ int lanes_or_zero(unsigned long long mask, int v) {
    int r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask)); return r;
}


template <typename OT>
__device__ __forceinline__ V4<OT> ld4(GP<OT> row, unsigned idx0) {
    return *(const V4<OT> __attribute__((address_space(1))) *)(row + idx0);
}
template <typename OT>
__device__ __forceinline__ void st4(GP<OT> row, unsigned idx0, V4<OT> v) {
    *(V4<OT> __attribute__((address_space(1))) *)(row + idx0) = v;
}
// streaming variants: rows that are (almost) never read again should not displace the rows that are
template <typename OT>
__device__ __forceinline__ void st4_nt(GP<OT> row, unsigned idx0, V4<OT> v) {
    __builtin_nontemporal_store(v, (V4<OT> __attribute__((address_space(1))) *)(row + idx0));
}
template <typename OT>
__device__ __forceinline__ V4<OT> ld4_nt(GP<OT> row, unsigned idx0) {
    return __builtin_nontemporal_load((const V4<OT> __attribute__((address_space(1))) *)(row + idx0));
}

// ---------------------------------------------------------------- CIGAR out
__device__ __forceinline__ void cig_append(GP<uint32_t> ops, uint32_t &cnt, uint32_t cap, int op,
                                           int len, int &err) {
    if (len <= 0) return;
    if (cnt > 0 && (int)(ops[cnt - 1] & 15u) == op) { ops[cnt - 1] += (uint32_t)len << 4; return; }
    if (cnt >= cap) { err |= SR_DEV_ERR_CIGAR_OVERFLOW; return; }
    ops[cnt++] = ((uint32_t)len << 4) | (uint32_t)op;
}

__device__ __forceinline__ void bt_best(int &bo, int &bty, int off, int type) {
    if (off < 0) return;
    if (off > bo || (off == bo && type > bty)) { bo = off; bty = type; }
}

template <int NT>
__device__ __forceinline__ void load_seq_lds(uint32_t *dst, GP<const uint32_t> src, int nwords_with_pad) {
    for (int i = threadIdx.x; i < nwords_with_pad; i += NT) dst[i] = src[i];
}

