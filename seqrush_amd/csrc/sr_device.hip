// sr_device.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// Kernel 1  sr_align_kernel : one sequence pair per workgroup (persistent
//   workgroups pulling pairs from a device queue).  Orientation scoring,
//   then exact gap-affine(-2p) end-to-end alignment in WFA2 "Ultralow"
//   (biWFA) mode: forward/reverse score-only wavefronts, breakpoint
//   detection, explicit recursion stack in LDS, plain WFA + backtrace for the
//   base cases.  Sequences are 2-bit packed and staged in LDS; one wavefront
//   diagonal per lane, neighbour diagonals k-1/k+1 are adjacent lanes' cells
//   of older ring levels.  Emits a run-length CIGAR per pair.
// Kernel 2  sr_unite_kernel : CIGAR -> exact-match runs (len >= k) -> per
//   base unite() on the uf_rush node array with device-scope atomicCAS and
//   path halving (BidirectedUnionFind::unite_matching_region).
// Kernel 3+ uf init, canonical min-Pos labels, label merge (multi-GPU).
//
// The alignment rules implemented here are exactly the ones written in
// oracle/wfa.c (restated WFA2-lib semantics); reference call sites:
// src/seqrush.rs:611-757, 1134-1481; src/bidirected_union_find.rs:60-98;
// uf_rush-0.2.1/src/lib.rs:112-208.  Integer DP and atomics only: no MFMA.
#include <hip/hip_runtime.h>
#include <limits.h>
#include "sr_internal.h"

#define WG SR_WG   // unite kernel workgroup size
#define NULLV SR_NULL_OFF
#define BT_TMP_CAP 2048

struct Seg { int pb, pe, tb, te; int cb, ce; int score_rem; };

struct Shared {
    int red_maxak[3][2];
    int reached[3][2];
    int mak[2][SR_MAX_SCOPE + 1];
    int cand[5];
    int bp_score, bp_score_f, bp_score_r, bp_k_f, bp_k_r, bp_comp, bp_off_f, bp_off_r;
    Seg stack[SR_STACK_DEPTH];
    int sp;
    uint32_t bt_tmp[BT_TMP_CAP];
    int bt_n;
    uint32_t cig_cnt;
    int pair;
    int err;
    int score_acc;
};

#define RFL(x) __builtin_amdgcn_readfirstlane(x)

// Explicit address spaces: pointers that travel inside a by-value kernel
// argument struct are generic ("flat") for hipcc; flat loads cost a VGPR pair
// per address and the slow path.  GP = global (HBM) pointer, LP = LDS pointer.
template <typename T> using GP = T __attribute__((address_space(1))) *;
typedef const uint32_t __attribute__((address_space(3))) *LP;

// One aligner's wavefront storage.  Modular (score-only) mode: M ring of
// scope+1 levels, "hot" I/D rings of e+2 levels (all the recurrences read),
// plus a "cold" I/D history of scope+1 levels that only the breakpoint
// detection reads (written with non-temporal stores so it does not displace
// the hot rings from L2 / Infinity Cache).  Full mode (base case): every level
// kept, hot == history.
template <typename OT>
struct Dir {
    GP<OT> m;        // M rows
    GP<OT> g[4];     // I1, I2, D1, D2 rows (index comp-1)
    GP<OT> cold;     // [slot][4][cap] or nullptr
    int nsm, nsg1, nsg2, nsc;
    int cap, shift;
    int modular;
    int rev;
    int pb, pe, tb, te;
    int plen, tlen;
    int begin;
};

template <typename OT>
__device__ __forceinline__ GP<OT> rowk(const Dir<OT> &d, int s, int comp) {
    if (comp == SR_C_M) return d.m + (size_t)(d.modular ? (s % d.nsm) : s) * (size_t)d.cap + d.shift;
    const int ns = (comp == SR_C_I1 || comp == SR_C_D1) ? d.nsg1 : d.nsg2;
    return d.g[comp - 1] + (size_t)(d.modular ? (s % ns) : s) * (size_t)d.cap + d.shift;
}
// same rows without the diagonal shift: indexed by the non-negative
// idx = k + shift so that hipcc can use SGPR-base + 32-bit VGPR-offset loads
template <typename OT>
__device__ __forceinline__ GP<OT> rowb(const Dir<OT> &d, int s, int comp) { return rowk(d, s, comp) - d.shift; }
template <typename OT>
__device__ __forceinline__ GP<OT> hrowb(const Dir<OT> &d, int s, int comp);
// row used by breakpoint detection / backtrace-independent history reads
template <typename OT>
__device__ __forceinline__ GP<OT> hrowk(const Dir<OT> &d, int s, int comp) {
    if (comp == SR_C_M || !d.cold) return rowk(d, s, comp);
    return d.cold + ((size_t)(s % d.nsc) * 4 + (comp - 1)) * (size_t)d.cap + d.shift;
}

template <typename OT>
__device__ __forceinline__ GP<OT> hrowb(const Dir<OT> &d, int s, int comp) { return hrowk(d, s, comp) - d.shift; }

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

// 16 bases starting at base i (2 bits each, base i in the low bits)
__device__ __forceinline__ uint32_t win_fwd(LP w, int i) {
    const int wi = i >> 4, sh = (i & 15) << 1;
    const uint64_t v = ((uint64_t)w[wi + 1] << 32) | (uint64_t)w[wi];
    return (uint32_t)(v >> sh);
}
// 16 bases ending at base i (base i in the high bits)
__device__ __forceinline__ uint32_t win_rev(LP w, int i) {
    return win_fwd(w, i - 15);
}

// number of equal bases walking forward from (pi, ti), at most n
__device__ __forceinline__ int ext_fwd(LP P, LP T, int pi, int ti, int n) {
    int tot = 0;
    while (tot < n) {
        const uint32_t x = win_fwd(P, pi + tot) ^ win_fwd(T, ti + tot);
        int c = x ? ((__ffs((int)x) - 1) >> 1) : 16;
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
        int c = x ? (__clz((int)x) >> 1) : 16;
        c = min(c, n - tot);
        tot += c;
        if (x) break;
    }
    return tot;
}

// One score step of one aligner: compute + bound + extend over the write
// range; reduction of the M max antidiagonal and the end test go to LDS slot
// [ev % 3][side].  Caller must __syncthreads() before reading them.
template <typename OT, bool TWO, int NT>
__device__ __forceinline__ void wf_step(const Dir<OT> &d, const SrPen &pen, int s,
                                        LP P, LP T, int check_comp,
                                        Shared &sh, int ev, int side, unsigned long long &cells) {
    const int tid = threadIdx.x;
    const int plen = d.plen, tlen = d.tlen;
    const int R = reach(pen, s, d.begin);
    const int klo = max(-plen, -R), khi = min(tlen, R);
    const int wlo = max(-plen - 1, -R - pen.scope - 1), whi = min(tlen + 1, R + pen.scope + 1);
    const int k_end = tlen - plen;
    const GP<OT> pMx = (s >= pen.x) ? rowb(d, s - pen.x, SR_C_M) : nullptr;
    const GP<OT> pMo1 = (s >= pen.o1 + pen.e1) ? rowb(d, s - pen.o1 - pen.e1, SR_C_M) : nullptr;
    const GP<OT> pI1 = (s >= pen.e1) ? rowb(d, s - pen.e1, SR_C_I1) : nullptr;
    const GP<OT> pD1 = (s >= pen.e1) ? rowb(d, s - pen.e1, SR_C_D1) : nullptr;
    GP<OT> pMo2 = nullptr, pI2 = nullptr, pD2 = nullptr;
    if (TWO) {
        if (s >= pen.o2 + pen.e2) pMo2 = rowb(d, s - pen.o2 - pen.e2, SR_C_M);
        if (s >= pen.e2) { pI2 = rowb(d, s - pen.e2, SR_C_I2); pD2 = rowb(d, s - pen.e2, SR_C_D2); }
    }
    GP<OT> oM = rowb(d, s, SR_C_M), oI1 = rowb(d, s, SR_C_I1), oD1 = rowb(d, s, SR_C_D1);
    GP<OT> oI2 = rowb(d, s, SR_C_I2), oD2 = rowb(d, s, SR_C_D2);
    GP<OT> cI1 = nullptr, cI2 = nullptr, cD1 = nullptr, cD2 = nullptr;
    if (d.cold) {
        cI1 = hrowb(d, s, SR_C_I1); cD1 = hrowb(d, s, SR_C_D1);
        cI2 = hrowb(d, s, SR_C_I2); cD2 = hrowb(d, s, SR_C_D2);
    }
    int my_ak = 0;
    bool my_reached = false;
    for (int k = wlo + tid; k <= whi; k += NT) {
        int m = NULLV, i1 = NULLV, i2 = NULLV, d1 = NULLV, d2 = NULLV;
        const unsigned idx = (unsigned)(k + d.shift);
        if (k >= klo && k <= khi) {
            if (s == 0) {
                if (k == 0) {
                    if (d.begin == SR_C_M) m = 0;
                    else if (d.begin == SR_C_I1) i1 = 0;
                    else if (d.begin == SR_C_I2) i2 = 0;
                    else if (d.begin == SR_C_D1) d1 = 0;
                    else d2 = 0;
                }
            } else {
                const unsigned lim = (unsigned)min(tlen, plen + k);
                {
                    const int a = pMo1 ? (int)pMo1[idx - 1] : NULLV;
                    const int b = pI1 ? (int)pI1[idx - 1] : NULLV;
                    i1 = bnd(max(a, b) + 1, lim);
                    const int c = pMo1 ? (int)pMo1[idx + 1] : NULLV;
                    const int e = pD1 ? (int)pD1[idx + 1] : NULLV;
                    d1 = bnd(max(c, e), lim);
                }
                if (TWO) {
                    const int a = pMo2 ? (int)pMo2[idx - 1] : NULLV;
                    const int b = pI2 ? (int)pI2[idx - 1] : NULLV;
                    i2 = bnd(max(a, b) + 1, lim);
                    const int c = pMo2 ? (int)pMo2[idx + 1] : NULLV;
                    const int e = pD2 ? (int)pD2[idx + 1] : NULLV;
                    d2 = bnd(max(c, e), lim);
                }
                const int mx = pMx ? (int)pMx[idx] : NULLV;
                m = bnd(mx + 1, lim);
                m = max(m, max(max(i1, i2), max(d1, d2)));
                if (m < 0) m = NULLV;
            }
            if (m >= 0) {
                const int v = m - k, h = m;
                const int n = min(plen - v, tlen - h);
                if (n > 0) {
                    int e;
                    if (!d.rev) e = ext_fwd(P, T, d.pb + v, d.tb + h, n);
                    else e = ext_rev(P, T, d.pe - 1 - v, d.te - 1 - h, n);
                    m += e;
                }
                my_ak = max(my_ak, 2 * m - k);
            }
            if (k == k_end) {
                int val = m;
                if (check_comp == SR_C_I1) val = i1;
                else if (check_comp == SR_C_I2) val = i2;
                else if (check_comp == SR_C_D1) val = d1;
                else if (check_comp == SR_C_D2) val = d2;
                if (val >= tlen) my_reached = true;
            }
        }
        oM[idx] = (OT)m; oI1[idx] = (OT)i1; oD1[idx] = (OT)d1;
        if (TWO) { oI2[idx] = (OT)i2; oD2[idx] = (OT)d2; }
        if (cI1) {
            __builtin_nontemporal_store((OT)i1, &cI1[idx]);
            __builtin_nontemporal_store((OT)d1, &cD1[idx]);
            if (TWO) {
                __builtin_nontemporal_store((OT)i2, &cI2[idx]);
                __builtin_nontemporal_store((OT)d2, &cD2[idx]);
            }
        }
    }
    my_ak = wave_max(my_ak);
    const int slot = ev % 3;
    if ((tid & 63) == 0 && my_ak > 0) atomicMax(&sh.red_maxak[slot][side], my_ak);
    if (my_reached) sh.reached[slot][side] = 1;
    if (tid == 0) {
        sh.red_maxak[(ev + 1) % 3][side] = 0;
        sh.reached[(ev + 1) % 3][side] = 0;
        if (khi >= klo) cells += (unsigned long long)(khi - klo + 1);
    }
}

// ---------------------------------------------------------------- CIGAR out
__device__ __forceinline__ void cig_append(GP<uint32_t> ops, uint32_t &cnt, uint32_t cap, int op,
                                           int len, int &err) {
    if (len <= 0) return;
    if (cnt > 0 && (int)(ops[cnt - 1] & 15u) == op) { ops[cnt - 1] += (uint32_t)len << 4; return; }
    if (cnt >= cap) { err |= SR_DEV_ERR_CIGAR_OVERFLOW; return; }
    ops[cnt++] = ((uint32_t)len << 4) | (uint32_t)op;
}

__device__ __forceinline__ void bt_push(Shared &sh, int op, int len, int &err) {
    if (len <= 0) return;
    int n = sh.bt_n;
    if (n > 0 && (int)(sh.bt_tmp[n - 1] & 15u) == op) { sh.bt_tmp[n - 1] += (uint32_t)len << 4; return; }
    if (n >= BT_TMP_CAP) { err |= SR_DEV_ERR_CIGAR_OVERFLOW; return; }
    sh.bt_tmp[n] = ((uint32_t)len << 4) | (uint32_t)op;
    sh.bt_n = n + 1;
}

__device__ __forceinline__ void bt_best(int &bo, int &bty, int off, int type) {
    if (off < 0) return;
    if (off > bo || (off == bo && type > bty)) { bo = off; bty = type; }
}

// thread-0 backtrace over the full history (oracle/wfa.c wfa_full)
template <typename OT, bool TWO>
__device__ __noinline__ int backtrace(const Dir<OT> d, const SrPen pen, int score, int cb, int ce,
                                      Shared &sh) {
    int err = 0;
    const int plen = d.plen, tlen = d.tlen;
    int s = score, k = tlen - plen, comp = ce, off = tlen;
    sh.bt_n = 0;
    for (int guard = 0; guard < 4 * (plen + tlen) + 64; guard++) {
        if (comp == SR_C_M) {
            if (s == 0) {
                if (cb != SR_C_M || k != 0) err |= SR_DEV_ERR_BACKTRACE;
                bt_push(sh, SR_OP_M, off, err);
                return err;
            }
            const unsigned lim = (unsigned)min(tlen, plen + k);
            int bo = NULLV, bty = 0;
            if (s >= pen.x) bt_best(bo, bty, bnd((int)rowk(d, s - pen.x, SR_C_M)[k] + 1, lim), 9);
            if (s >= pen.o1 + pen.e1) {
                const GP<OT> r = rowk(d, s - pen.o1 - pen.e1, SR_C_M);
                bt_best(bo, bty, bnd((int)r[k - 1] + 1, lim), 1);
                bt_best(bo, bty, bnd((int)r[k + 1], lim), 5);
            }
            if (s >= pen.e1) {
                bt_best(bo, bty, bnd((int)rowk(d, s - pen.e1, SR_C_I1)[k - 1] + 1, lim), 2);
                bt_best(bo, bty, bnd((int)rowk(d, s - pen.e1, SR_C_D1)[k + 1], lim), 6);
            }
            if (TWO) {
                if (s >= pen.o2 + pen.e2) {
                    const GP<OT> r = rowk(d, s - pen.o2 - pen.e2, SR_C_M);
                    bt_best(bo, bty, bnd((int)r[k - 1] + 1, lim), 3);
                    bt_best(bo, bty, bnd((int)r[k + 1], lim), 7);
                }
                if (s >= pen.e2) {
                    bt_best(bo, bty, bnd((int)rowk(d, s - pen.e2, SR_C_I2)[k - 1] + 1, lim), 4);
                    bt_best(bo, bty, bnd((int)rowk(d, s - pen.e2, SR_C_D2)[k + 1], lim), 8);
                }
            }
            if (bty == 0 || bo > off) { err |= SR_DEV_ERR_BACKTRACE; return err; }
            bt_push(sh, SR_OP_M, off - bo, err);
            off = bo;
            switch (bty) {
            case 9: bt_push(sh, SR_OP_X, 1, err); off -= 1; s -= pen.x; break;
            case 1: bt_push(sh, SR_OP_I, 1, err); off -= 1; k -= 1; s -= pen.o1 + pen.e1; break;
            case 2: bt_push(sh, SR_OP_I, 1, err); off -= 1; k -= 1; s -= pen.e1; comp = SR_C_I1; break;
            case 3: bt_push(sh, SR_OP_I, 1, err); off -= 1; k -= 1; s -= pen.o2 + pen.e2; break;
            case 4: bt_push(sh, SR_OP_I, 1, err); off -= 1; k -= 1; s -= pen.e2; comp = SR_C_I2; break;
            case 5: bt_push(sh, SR_OP_D, 1, err); k += 1; s -= pen.o1 + pen.e1; break;
            case 6: bt_push(sh, SR_OP_D, 1, err); k += 1; s -= pen.e1; comp = SR_C_D1; break;
            case 7: bt_push(sh, SR_OP_D, 1, err); k += 1; s -= pen.o2 + pen.e2; break;
            default: bt_push(sh, SR_OP_D, 1, err); k += 1; s -= pen.e2; comp = SR_C_D2; break;
            }
        } else {
            if (s == 0) {
                if (comp != cb || k != 0 || off != 0) err |= SR_DEV_ERR_BACKTRACE;
                return err;
            }
            const bool is_ins = (comp == SR_C_I1 || comp == SR_C_I2);
            const bool p1 = (comp == SR_C_I1 || comp == SR_C_D1);
            const int o = p1 ? pen.o1 : pen.o2, e = p1 ? pen.e1 : pen.e2;
            const unsigned lim = (unsigned)min(tlen, plen + k);
            int c_open = NULLV, c_ext = NULLV;
            if (is_ins) {
                if (s >= o + e) c_open = bnd((int)rowk(d, s - o - e, SR_C_M)[k - 1] + 1, lim);
                if (s >= e) c_ext = bnd((int)rowk(d, s - e, comp)[k - 1] + 1, lim);
            } else {
                if (s >= o + e) c_open = bnd((int)rowk(d, s - o - e, SR_C_M)[k + 1], lim);
                if (s >= e) c_ext = bnd((int)rowk(d, s - e, comp)[k + 1], lim);
            }
            bool take_ext;
            if (c_ext >= 0 && c_ext >= c_open) take_ext = true;
            else if (c_open >= 0) take_ext = false;
            else { err |= SR_DEV_ERR_BACKTRACE; return err; }
            if ((take_ext ? c_ext : c_open) != off) { err |= SR_DEV_ERR_BACKTRACE; return err; }
            if (is_ins) { bt_push(sh, SR_OP_I, 1, err); off -= 1; k -= 1; }
            else { bt_push(sh, SR_OP_D, 1, err); k += 1; }
            if (take_ext) s -= e; else { s -= o + e; comp = SR_C_M; }
        }
        if (s < 0) { err |= SR_DEV_ERR_BACKTRACE; return err; }
    }
    err |= SR_DEV_ERR_BACKTRACE;
    return err;
}

template <typename OT>
__device__ __forceinline__ void dir_seg(Dir<OT> &d, const Seg &sg) {
    d.pb = sg.pb; d.pe = sg.pe; d.tb = sg.tb; d.te = sg.te;
    d.plen = sg.pe - sg.pb; d.tlen = sg.te - sg.tb;
}

// plain WFA with full history + backtrace on one segment (all threads)
template <typename OT, bool TWO, int NT>
__device__ __forceinline__ void wfa_base(const Seg &sg, const SrAlignArgs &a, const SrPen &pen, GP<OT> hist,
                         LP P, LP T, Shared &sh, int &ev,
                         unsigned long long &cells, unsigned long long &steps, GP<uint32_t> ops,
                         uint32_t cap) {
    Dir<OT> d;
    const size_t comp_stride = (size_t)a.hist_levels * (size_t)a.hist_w;
    d.m = hist;
    for (int c = 0; c < 4; c++) d.g[c] = hist + (size_t)(c + 1) * comp_stride;
    d.cold = nullptr; d.nsm = d.nsg1 = d.nsg2 = d.nsc = a.hist_levels;
    d.cap = a.hist_w; d.shift = a.hist_w / 2; d.modular = 0; d.rev = 0; d.begin = sg.cb;
    dir_seg(d, sg);
    int s = 0;
    bool found = false;
    for (;;) {
        wf_step<OT, TWO, NT>(d, pen, s, P, T, sg.ce, sh, ev, 0, cells);
        __syncthreads();
        const bool r = sh.reached[ev % 3][0] != 0;
        ev++;
        steps++;
        if (r) { found = true; break; }
        if (s + 1 >= a.hist_levels) break;
        s++;
    }
    if (threadIdx.x == 0) {
        int err = 0;
        if (!found) err |= SR_DEV_ERR_BASE_OVERFLOW;
        else {
            err |= backtrace<OT, TWO>(d, pen, s, sg.cb, sg.ce, sh);
            uint32_t cnt = sh.cig_cnt;
            for (int i = sh.bt_n - 1; i >= 0; i--)
                cig_append(ops, cnt, cap, (int)(sh.bt_tmp[i] & 15u), (int)(sh.bt_tmp[i] >> 4), err);
            sh.cig_cnt = cnt;
        }
        if (err) sh.err |= err;
    }
    __syncthreads();
}

template <typename OT>
__device__ __forceinline__ void bp_try(const Dir<OT> &d0, const Dir<OT> &d1, int c, int gap,
                                       int score_0, int score_i, int kinv, bool bp_forward, Shared &sh) {
    const int k0 = sh.cand[c];
    if (k0 == INT_MAX) return;
    if (!(score_0 + score_i - gap < sh.bp_score)) return;
    const int k1 = kinv - k0;
    const int o0 = (int)hrowk(d0, score_0, c)[k0];
    const int o1 = (int)hrowk(d1, score_i, c)[k1];
    if (bp_forward) {
        sh.bp_score_f = score_0; sh.bp_score_r = score_i;
        sh.bp_k_f = k0; sh.bp_k_r = k1; sh.bp_off_f = o0; sh.bp_off_r = o1;
    } else {
        sh.bp_score_f = score_i; sh.bp_score_r = score_0;
        sh.bp_k_f = k1; sh.bp_k_r = k0; sh.bp_off_f = o1; sh.bp_off_r = o0;
    }
    sh.bp_score = score_0 + score_i - gap;
    sh.bp_comp = c;
}

// breakpoint detection between level score_0 of aligner a0 and the last
// `scope` levels of aligner a1 (oracle/wfa.c bialign_overlap)
template <typename OT, bool TWO, int NT>
__device__ __forceinline__ void bi_overlap(const Dir<OT> &d0, const Dir<OT> &d1, int a0, int a1, const SrPen &pen,
                           int score_0, int score_1, bool bp_forward, Shared &sh) {
    const int tid = threadIdx.x;
    const int plen = d0.plen, tlen = d0.tlen;
    const int kinv = tlen - plen;
    __syncthreads();                                      // mak[] / bp_score of the last step visible
    const int mak0 = RFL(sh.mak[a0][score_0 % d0.nsm]);
    const int gapmax = TWO ? max(pen.o1, pen.o2) : pen.o1;
    const int R0 = reach(pen, score_0, d0.begin);
    const int klo0 = max(-plen, -R0), khi0 = min(tlen, R0);
    for (int i = 0; i < pen.scope; i++) {
        const int score_i = score_1 - i;
        if (score_i < 0) break;
        const int mak1 = RFL(sh.mak[a1][score_i % d1.nsm]);
        if (mak0 + mak1 < plen + tlen) continue;            // no diagonal can overlap
        if (score_0 + score_i - gapmax >= RFL(sh.bp_score)) continue;
        if (tid < 5) sh.cand[tid] = INT_MAX;
        __syncthreads();
        const int R1 = reach(pen, score_i, d1.begin);
        const int klo1 = max(-plen, -R1), khi1 = min(tlen, R1);
        const int lo = max(klo0, kinv - khi1), hi = min(khi0, kinv - klo1);
        for (int k0 = lo + tid; k0 <= hi; k0 += NT) {
            const int k1 = kinv - k0;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                if (!TWO && (c == SR_C_I2 || c == SR_C_D2)) continue;
                const int o0 = (int)hrowk(d0, score_0, c)[k0];
                const int o1 = (int)hrowk(d1, score_i, c)[k1];
                if (o0 >= 0 && o1 >= 0 && o0 + o1 >= tlen) atomicMin(&sh.cand[c], k0);
            }
        }
        __syncthreads();
        if (tid == 0) {
            // order: D2, I2, D1, I1, M with the running thresholds
            if (TWO) {
                bp_try(d0, d1, SR_C_D2, pen.o2, score_0, score_i, kinv, bp_forward, sh);
                bp_try(d0, d1, SR_C_I2, pen.o2, score_0, score_i, kinv, bp_forward, sh);
            }
            bp_try(d0, d1, SR_C_D1, pen.o1, score_0, score_i, kinv, bp_forward, sh);
            bp_try(d0, d1, SR_C_I1, pen.o1, score_0, score_i, kinv, bp_forward, sh);
            bp_try(d0, d1, SR_C_M, 0, score_0, score_i, kinv, bp_forward, sh);
        }
        __syncthreads();
    }
}

template <typename OT>
__device__ __forceinline__ void dir_ring(Dir<OT> &d, GP<OT> base, const SrAlignArgs &a, const SrPen &pen) {
    // layout of one direction: M[(scope_max+1)] | I1 I2 D1 D2 hot [(emax+2)] each | cold [(scope_max+1)][4]
    const size_t cap = (size_t)a.ring_cap;
    d.m = base;
    GP<OT> p = base + (size_t)(a.ring_scope + 1) * cap;
    for (int c = 0; c < 4; c++) { d.g[c] = p; p += (size_t)a.ring_hot * cap; }
    d.cold = p;
    d.nsm = pen.scope + 1; d.nsg1 = pen.e1 + 2; d.nsg2 = pen.two ? pen.e2 + 2 : 2; d.nsc = pen.scope + 1;
    d.cap = a.ring_cap; d.modular = 1;
}

// computes F level done_f+1 and/or R level done_r+1 in one pass (one barrier)
#define STEP_BOTH(DO_F, DO_R)                                                                      \
    do {                                                                                           \
        const bool do_f_ = (DO_F), do_r_ = (DO_R);                                                 \
        if (do_f_) wf_step<OT, TWO, NT>(F, pen, done_f + 1, P, T, -1, sh, ev, 0, cells);           \
        if (do_r_) wf_step<OT, TWO, NT>(R, pen, done_r + 1, P, T, -1, sh, ev, 1, cells);           \
        __syncthreads();                                                                           \
        if (do_f_) {                                                                               \
            done_f++; ak_f_done = RFL(sh.red_maxak[ev % 3][0]);                                    \
            if (tid == 0) sh.mak[0][done_f % F.nsm] = ak_f_done;                                   \
            steps++;                                                                               \
        }                                                                                          \
        if (do_r_) {                                                                               \
            done_r++; ak_r_done = RFL(sh.red_maxak[ev % 3][1]);                                    \
            if (tid == 0) sh.mak[1][done_r % R.nsm] = ak_r_done;                                   \
            steps++;                                                                               \
        }                                                                                          \
        ev++;                                                                                      \
    } while (0)

// biWFA breakpoint search on one segment (oracle/wfa.c bialign_find_breakpoint).
// Forward and reverse aligners are stepped together: the level the other
// aligner will need next is computed speculatively in the same pass (it does
// not depend on this one), so one barrier serves two wavefront steps.
template <typename OT, bool TWO, int NT>
__device__ __forceinline__ bool find_breakpoint(const Seg &sg, const SrAlignArgs &a, const SrPen &pen, GP<OT> ring,
                                LP P, LP T, Shared &sh, int &ev,
                                unsigned long long &cells, unsigned long long &steps) {
    const int tid = threadIdx.x;
    Dir<OT> F, R;
    dir_ring(F, ring, a, pen);
    dir_ring(R, ring + a.ring_dir_stride, a, pen);
    dir_seg(F, sg); dir_seg(R, sg);
    F.rev = 0; R.rev = 1; F.begin = sg.cb; R.begin = sg.ce;
    F.shift = F.plen + 1; R.shift = R.plen + 1;
    const int plen = F.plen, tlen = F.tlen;
    const int max_antidiagonal = plen + tlen - 1;
    const int scope = pen.scope;
    const int gap_opening = TWO ? max(pen.o1, pen.o2) : pen.o1;
    const long long smax = 2LL * ((long long)pen.o1 * 2 + (long long)pen.e1 * (plen + tlen)) + 1024;
    int score_f = 0, score_r = 0;
    int done_f = -1, done_r = -1;        // highest level computed
    int ak_f_done = 0, ak_r_done = 0;    // max antidiagonal of level done_f / done_r
    if (tid == 0) sh.bp_score = INT_MAX;
    STEP_BOTH(true, true);
    int f_max_ak = ak_f_done, r_max_ak = ak_r_done;
    bool last_wf_forward = false;
    bool ok = true;
    for (;;) {
        if (f_max_ak + r_max_ak >= max_antidiagonal) break;
        ++score_f;
        if (done_f < score_f) STEP_BOTH(true, done_r < score_r + 1);
        f_max_ak = ak_f_done;
        last_wf_forward = true;
        if (f_max_ak + r_max_ak >= max_antidiagonal) break;
        ++score_r;
        if (done_r < score_r) STEP_BOTH(done_f < score_f + 1, true);
        r_max_ak = ak_r_done;
        last_wf_forward = false;
        if ((long long)score_f + score_r > smax) { ok = false; break; }
    }
    while (ok) {
        if (last_wf_forward) {
            const int min_score_r = (score_r > scope - 1) ? score_r - (scope - 1) : 0;
            __syncthreads();
            if (score_f + min_score_r - gap_opening >= RFL(sh.bp_score)) break;
            bi_overlap<OT, TWO, NT>(F, R, 0, 1, pen, score_f, score_r, true, sh);
            ++score_r;
            if (done_r < score_r) STEP_BOTH(done_f < score_f + 1, true);
        }
        const int min_score_f = (score_f > scope - 1) ? score_f - (scope - 1) : 0;
        __syncthreads();
        if (min_score_f + score_r - gap_opening >= RFL(sh.bp_score)) break;
        bi_overlap<OT, TWO, NT>(R, F, 1, 0, pen, score_r, score_f, false, sh);
        ++score_f;
        if (done_f < score_f) STEP_BOTH(true, done_r < score_r + 1);
        last_wf_forward = true;
        if ((long long)score_f + score_r > smax) { ok = false; break; }
    }
    __syncthreads();
    if (ok && RFL(sh.bp_score) == INT_MAX) ok = false;
    if (!ok && tid == 0) sh.err |= SR_DEV_ERR_SCORE_BOUND;
    __syncthreads();
    return ok;
}

// score-only end-to-end WFA (orientation check); returns INT_MAX when the
// score exceeds max_score (max_score < 0: unbounded)
template <typename OT, int NT>
__device__ __forceinline__ int wfa_score_only(int plen, int tlen, const SrAlignArgs &a, const SrPen &pen, GP<OT> ring,
                              LP P, LP T, int max_score, Shared &sh,
                              int &ev, unsigned long long &cells, unsigned long long &steps) {
    Dir<OT> d;
    dir_ring(d, ring, a, pen);
    d.cold = nullptr;                      // no breakpoint detection here: no history
    d.rev = 0;
    d.pb = 0; d.pe = plen; d.tb = 0; d.te = tlen; d.plen = plen; d.tlen = tlen;
    d.shift = plen + 1; d.begin = SR_C_M;
    const long long smax = (long long)pen.o1 * 2 + (long long)pen.e1 * (plen + tlen) + 64;
    int s = 0, res = INT_MAX;
    for (;;) {
        wf_step<OT, false, NT>(d, pen, s, P, T, SR_C_M, sh, ev, 0, cells);
        __syncthreads();
        const bool r = sh.reached[ev % 3][0] != 0;
        ev++; steps++;
        if (r) { res = s; break; }
        if (s > smax) { if (threadIdx.x == 0) sh.err |= SR_DEV_ERR_SCORE_BOUND; break; }
        if (max_score >= 0 && s >= max_score) break;
        s++;
    }
    __syncthreads();
    return res;
}

template <int NT>
__device__ __forceinline__ void load_seq_lds(uint32_t *dst, GP<const uint32_t> src, int nwords_with_pad) {
    for (int i = threadIdx.x; i < nwords_with_pad; i += NT) dst[i] = src[i];
}

template <typename OT, int NT, bool TWO>
__global__ void __launch_bounds__(NT) sr_align_kernel(SrAlignArgs a) {
    extern __shared__ uint32_t lds_seq[];     // 3 regions of max_words: P fwd, P rc, T
    __shared__ Shared sh;
    const int tid = threadIdx.x;
    GP<OT> ring = (GP<OT>)(OT *)a.ring + (size_t)blockIdx.x * a.ring_wg_stride;
    GP<OT> hist = (GP<OT>)(OT *)a.hist + (size_t)blockIdx.x * a.hist_wg_stride;
    unsigned long long cells = 0, steps = 0, nbase = 0, nbp = 0;
    int ev = 0;
    for (;;) {
        if (tid == 0) {
            sh.pair = (int)atomicAdd(a.queue_head, 1u);
            sh.err = 0; sh.sp = 0; sh.cig_cnt = 0; sh.score_acc = 0;
            for (int i = 0; i < 3; i++) { sh.red_maxak[i][0] = sh.red_maxak[i][1] = 0; sh.reached[i][0] = sh.reached[i][1] = 0; }
        }
        __syncthreads();
        const int pair = RFL(sh.pair);
        if (pair >= (int)a.npairs) break;
        const uint32_t q = a.pair_q[pair], t = a.pair_t[pair];
        const int plen = (int)a.seqlen[q], tlen = (int)a.seqlen[t];
        const int pw = ((plen + 15) >> 4) + 2, tw = ((tlen + 15) >> 4) + 2;
        uint32_t *Pf = lds_seq, *Pr = lds_seq + a.max_words, *Tt = lds_seq + 2 * (size_t)a.max_words;
        load_seq_lds<NT>(Pf, (GP<const uint32_t>)a.seqwords + a.word_off_fwd[q] - 1, pw);
        load_seq_lds<NT>(Pr, (GP<const uint32_t>)a.seqwords + a.word_off_rc[q] - 1, pw);
        load_seq_lds<NT>(Tt, (GP<const uint32_t>)a.seqwords + a.word_off_fwd[t] - 1, tw);
        __syncthreads();
        const LP T = (LP)(Tt + 1);
        // ---- orientation (forward on ties; reverse scored only up to fwd-1)
        const int fwd = wfa_score_only<OT, NT>(plen, tlen, a, a.ori, ring, (LP)(Pf + 1), T, -1, sh, ev, cells, steps);
        int rev = INT_MAX;
        bool is_rev = false;
        if (fwd > 0 && fwd != INT_MAX) {
            rev = wfa_score_only<OT, NT>(plen, tlen, a, a.ori, ring, (LP)(Pr + 1), T, fwd - 1, sh, ev, cells, steps);
            is_rev = rev < fwd;
        }
        const LP P = (LP)((is_rev ? Pr : Pf) + 1);
        GP<uint32_t> ops = (GP<uint32_t>)a.cigar_ops + a.cigar_base[pair];
        const uint32_t cap = (uint32_t)(a.cigar_base[pair + 1] - a.cigar_base[pair]);
        // ---- main alignment
        if (tid == 0) {
            Seg s0; s0.pb = 0; s0.pe = plen; s0.tb = 0; s0.te = tlen; s0.cb = SR_C_M; s0.ce = SR_C_M;
            s0.score_rem = (a.mem_mode == 3) ? INT_MAX : -1;   // -1: force plain WFA
            sh.stack[0] = s0; sh.sp = 1;
        }
        __syncthreads();
        while (RFL(sh.sp) > 0 && RFL(sh.err) == 0) {
            const int top = RFL(sh.sp) - 1;
            Seg sg;
            sg.pb = RFL(sh.stack[top].pb); sg.pe = RFL(sh.stack[top].pe);
            sg.tb = RFL(sh.stack[top].tb); sg.te = RFL(sh.stack[top].te);
            sg.cb = RFL(sh.stack[top].cb); sg.ce = RFL(sh.stack[top].ce);
            sg.score_rem = RFL(sh.stack[top].score_rem);
            __syncthreads();
            if (tid == 0) sh.sp = top;
            const int sp_len = sg.pe - sg.pb, st_len = sg.te - sg.tb;
            if (st_len == 0 || sp_len == 0) {
                if (tid == 0) {
                    int err = 0; uint32_t cnt = sh.cig_cnt;
                    if (st_len == 0) cig_append(ops, cnt, cap, SR_OP_D, sp_len, err);
                    else cig_append(ops, cnt, cap, SR_OP_I, st_len, err);
                    sh.cig_cnt = cnt; if (err) sh.err |= err;
                }
                __syncthreads();
                continue;
            }
            const bool base = sg.score_rem <= 250 || max(sp_len, st_len) <= 100;
            if (base) {
                nbase++;
                wfa_base<OT, TWO, NT>(sg, a, a.pen, hist, P, T, sh, ev, cells, steps, ops, cap);
                continue;
            }
            nbp++;
            const bool ok = find_breakpoint<OT, TWO, NT>(sg, a, a.pen, ring, P, T, sh, ev, cells, steps);
            if (ok && tid == 0) {
                const int bh = sh.bp_off_f, bv = sh.bp_off_f - sh.bp_k_f;
                if (bv < 0 || bv > sp_len || bh < 0 || bh > st_len) sh.err |= SR_DEV_ERR_BREAKPOINT;
                else if (sh.sp + 2 > SR_STACK_DEPTH) sh.err |= SR_DEV_ERR_STACK;
                else {
                    Seg h1; h1.pb = sg.pb + bv; h1.pe = sg.pe; h1.tb = sg.tb + bh; h1.te = sg.te;
                    h1.cb = sh.bp_comp; h1.ce = sg.ce; h1.score_rem = sh.bp_score_r;
                    Seg h0; h0.pb = sg.pb; h0.pe = sg.pb + bv; h0.tb = sg.tb; h0.te = sg.tb + bh;
                    h0.cb = sg.cb; h0.ce = sh.bp_comp; h0.score_rem = sh.bp_score_f;
                    sh.stack[sh.sp] = h1; sh.stack[sh.sp + 1] = h0; sh.sp += 2;
                }
            }
            __syncthreads();
        }
        __syncthreads();
        // ---- score of the final CIGAR (gaps costed once per merged run)
        const uint32_t cnt = RFL(sh.cig_cnt);
        int part = 0;
        for (uint32_t i = tid; i < cnt; i += NT) {
            const uint32_t op = ops[i] & 15u; const int len = (int)(ops[i] >> 4);
            if (op == SR_OP_X) part += len * a.pen.x;
            else if (op == SR_OP_I || op == SR_OP_D) {
                int g = a.pen.o1 + a.pen.e1 * len;
                if (TWO) g = min(g, a.pen.o2 + a.pen.e2 * len);
                part += g;
            }
        }
        if (part) atomicAdd(&sh.score_acc, part);
        __syncthreads();
        if (tid == 0) {
            a.is_reverse[pair] = is_rev ? 1 : 0;
            a.score[pair] = sh.err ? -1 : sh.score_acc;
            a.ori_fwd[pair] = fwd; a.ori_rev[pair] = rev;
            a.cigar_cnt[pair] = sh.err ? 0u : cnt;
            if (sh.err) atomicOr(a.error_flag, sh.err);
        }
        __syncthreads();
    }
    if (tid == 0) {
        atomicAdd(&a.counters[0], cells);
        atomicAdd(&a.counters[1], steps);
        atomicAdd(&a.counters[2], nbase);
        atomicAdd(&a.counters[3], nbp);
    }
}

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
    __shared__ unsigned sq[WG], st[WG], sm[WG];   // inclusive scans of one chunk
    __shared__ unsigned long long carry_q, carry_t;
    __shared__ unsigned wsum[3][WG / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long united = 0, runs = 0;
    int err = 0;
    for (uint32_t pair = blockIdx.x; pair < a.npairs; pair += gridDim.x) {
        const uint32_t cnt = a.cigar_cnt[pair];
        if (a.score[pair] < 0 || a.score[pair] > a.max_score[pair]) continue;   // uniform
        const uint32_t q = a.pair_q[pair], t = a.pair_t[pair];
        const unsigned long long qoff = a.seq_goff[q], toff = a.seq_goff[t];
        const unsigned long long qlen = a.seqlen[q];
        const bool rc = a.is_reverse[pair] != 0;
        const uint32_t *ops = a.cigar_ops + a.cigar_base[pair];
        if (tid == 0) { carry_q = 0; carry_t = 0; }
        __syncthreads();
        for (uint32_t base = 0; base < cnt; base += WG) {
            const uint32_t i = base + tid;
            unsigned dq = 0, dt = 0, ml = 0;
            if (i < cnt) {
                const uint32_t op = ops[i] & 15u; const unsigned len = ops[i] >> 4;
                if (op == SR_OP_M) { dq = len; dt = len; if ((unsigned long long)len >= a.min_match_len) ml = len; }
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
            const unsigned total_m = sm[WG - 1];
            const unsigned long long cq = carry_q, ct = carry_t;
            if (ml) runs++;
            for (unsigned j = tid; j < total_m; j += WG) {
                // op index: first idx with sm[idx] > j
                int lo = 0, hi = WG - 1;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (sm[mid] > j) hi = mid; else lo = mid + 1; }
                const unsigned len = ops[base + lo] >> 4;
                const unsigned within = j - (sm[lo] - len);
                const unsigned long long qpos = cq + (sq[lo] - len) + within;   // query-space index
                const unsigned long long tpos = ct + (st[lo] - len) + within;
                unsigned long long p1, p2 = (toff + tpos) << 1;
                if (rc) p1 = ((qoff + (qlen - 1 - qpos)) << 1) | 1ULL;
                else p1 = (qoff + qpos) << 1;
                if (p1 != p2) uf_unite(a.nodes, p1, p2, err);
                united++;
            }
            __syncthreads();
            if (tid == 0) { carry_q = cq + sq[WG - 1]; carry_t = ct + st[WG - 1]; }
            __syncthreads();
        }
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

// ------------------------------------------------------------------ launchers
extern "C" int srk_align_max_lds(void) { return 160 * 1024 - (int)sizeof(Shared) - 1024; }

template <typename OT, int NT, bool TWO>
static int launch_align3(const SrAlignArgs *a, int nwg, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 32 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)sr_align_kernel<OT, NT, TWO>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((sr_align_kernel<OT, NT, TWO>), dim3(nwg), dim3(NT), lds_bytes, st, *a);
    return (int)hipGetLastError();
}
template <typename OT, int NT>
static int launch_align(const SrAlignArgs *a, int nwg, size_t lds_bytes, hipStream_t st) {
    return a->pen.two ? launch_align3<OT, NT, true>(a, nwg, lds_bytes, st)
                      : launch_align3<OT, NT, false>(a, nwg, lds_bytes, st);
}

extern "C" int srk_align(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (off16) {
        if (nthreads == 512) return launch_align<int16_t, 512>(a, nwg, lds_bytes, st);
        return launch_align<int16_t, 256>(a, nwg, lds_bytes, st);
    }
    if (nthreads == 512) return launch_align<int32_t, 512>(a, nwg, lds_bytes, st);
    return launch_align<int32_t, 256>(a, nwg, lds_bytes, st);
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
