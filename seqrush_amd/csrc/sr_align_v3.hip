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
#include "sr_dev_common.h"

#define BT_TMP_CAP 1024

struct Seg { int pb, pe, tb, te; int cb, ce; int score_rem; };

// Uniform description of one aligner, kept in LDS so that the noinline step /
// overlap / backtrace functions have register pressure of their own.
struct DirL {
    unsigned long long m, g[4], cold, nullrow;   // global addresses (0 = none)
    int nsm, nsg1, nsg2, nsc;
    int cap, shift, modular, rev;
    int pb, pe, tb, te, plen, tlen, begin;
    // last computed level and its ring slots, double-buffered by step parity: a step reads
    // copy [ev & 1] while thread 0 writes copy [(ev + 1) & 1] (no wave can see a half update)
    int lvl[2], sm[2], sg1[2], sg2[2], sc[2];
};

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
    DirL dl[2];
    SrPen pen[2];            // [0] alignment penalties, [1] orientation penalties
    int offP, offT;          // word offsets of the pattern / text inside lds_seq
    unsigned long long cells;
};

__shared__ Shared g_sh;

// Register copy of a DirL.  Modular (score-only) mode: M ring of scope+1
// levels, "hot" I/D rings of e+2 levels (all the recurrences read), plus a
// "cold" I/D history of scope+1 levels that only breakpoint detection reads.
// Full mode (base case): every level kept, hot == history.
template <typename OT>
struct Dir {
    GP<OT> m, g[4], cold, nullrow;
    int nsm, nsg1, nsg2, nsc;
    int cap, shift, modular, rev;
    int pb, pe, tb, te, plen, tlen, begin;
    int lvl, sm, sg1, sg2, sc;
};

template <typename OT>
__device__ __forceinline__ Dir<OT> load_dir(int i, int par = 0) {
    const DirL &l = g_sh.dl[i];
    Dir<OT> d;
    d.m = (GP<OT>)rfl64(l.m);
    for (int c = 0; c < 4; c++) d.g[c] = (GP<OT>)rfl64(l.g[c]);
    d.cold = (GP<OT>)rfl64(l.cold); d.nullrow = (GP<OT>)rfl64(l.nullrow);
    d.nsm = RFL(l.nsm); d.nsg1 = RFL(l.nsg1); d.nsg2 = RFL(l.nsg2); d.nsc = RFL(l.nsc);
    d.cap = RFL(l.cap); d.shift = RFL(l.shift); d.modular = RFL(l.modular); d.rev = RFL(l.rev);
    d.pb = RFL(l.pb); d.pe = RFL(l.pe); d.tb = RFL(l.tb); d.te = RFL(l.te);
    d.plen = RFL(l.plen); d.tlen = RFL(l.tlen); d.begin = RFL(l.begin);
    d.lvl = RFL(l.lvl[par]); d.sm = RFL(l.sm[par]); d.sg1 = RFL(l.sg1[par]); d.sg2 = RFL(l.sg2[par]); d.sc = RFL(l.sc[par]);
    return d;
}
__device__ __forceinline__ SrPen load_pen(int sel) {
    const SrPen &q = g_sh.pen[sel];
    SrPen p;
    p.x = RFL(q.x); p.o1 = RFL(q.o1); p.e1 = RFL(q.e1); p.o2 = RFL(q.o2); p.e2 = RFL(q.e2);
    p.two = RFL(q.two); p.scope = RFL(q.scope);
    return p;
}

__device__ __forceinline__ int slot_inc(int v, int n) { return (v + 1 == n) ? 0 : v + 1; }
__device__ __forceinline__ int slot_back(int v, int delta, int n) { const int w = v - delta; return w < 0 ? w + n : w; }

// thread 0: publish the running state for the next step (parity np = (ev + 1) & 1);
// `stepped` = level lvl+1 of aligner i has just been computed
__device__ __forceinline__ void dirl_advance(int i, int par, bool stepped) {
    DirL &l = g_sh.dl[i];
    const int np = par ^ 1;
    if (stepped) {
        l.lvl[np] = l.lvl[par] + 1;
        if (l.modular) {
            l.sm[np] = slot_inc(l.sm[par], l.nsm); l.sg1[np] = slot_inc(l.sg1[par], l.nsg1);
            l.sg2[np] = slot_inc(l.sg2[par], l.nsg2); l.sc[np] = slot_inc(l.sc[par], l.nsc);
        } else { l.sm[np] = l.sm[par]; l.sg1[np] = l.sg1[par]; l.sg2[np] = l.sg2[par]; l.sc[np] = l.sc[par]; }
    } else {
        l.lvl[np] = l.lvl[par]; l.sm[np] = l.sm[par]; l.sg1[np] = l.sg1[par]; l.sg2[np] = l.sg2[par]; l.sc[np] = l.sc[par];
    }
}
__device__ __forceinline__ void dirl_reset(DirL &l) {
    for (int q = 0; q < 2; q++) { l.lvl[q] = -1; l.sm[q] = l.nsm - 1; l.sg1[q] = l.nsg1 - 1; l.sg2[q] = l.nsg2 - 1; l.sc[q] = l.nsc - 1; }
}

// unshifted row of level (lvl + 1 - delta) without integer division (hot path)
template <typename OT>
__device__ __forceinline__ GP<OT> rowq(const Dir<OT> &d, int delta, int comp) {
    int slot;
    if (comp == SR_C_M) {
        slot = d.modular ? slot_back(slot_inc(d.sm, d.nsm), delta, d.nsm) : d.lvl + 1 - delta;
        return d.m + (size_t)slot * (size_t)d.cap;
    }
    if (comp == SR_C_I1 || comp == SR_C_D1) slot = d.modular ? slot_back(slot_inc(d.sg1, d.nsg1), delta, d.nsg1) : d.lvl + 1 - delta;
    else slot = d.modular ? slot_back(slot_inc(d.sg2, d.nsg2), delta, d.nsg2) : d.lvl + 1 - delta;
    return d.g[comp - 1] + (size_t)slot * (size_t)d.cap;
}
template <typename OT>
__device__ __forceinline__ GP<OT> rowqc(const Dir<OT> &d, int comp) {   // cold history row of level lvl + 1
    return d.cold + ((size_t)slot_inc(d.sc, d.nsc) * 4 + (comp - 1)) * (size_t)d.cap;
}
// arbitrary level (cold paths: breakpoint detection, backtrace), shifted: index by k
template <typename OT>
__device__ __forceinline__ GP<OT> rowk(const Dir<OT> &d, int s, int comp) {
    if (comp == SR_C_M) return d.m + (size_t)(d.modular ? (s % d.nsm) : s) * (size_t)d.cap + d.shift;
    const int ns = (comp == SR_C_I1 || comp == SR_C_D1) ? d.nsg1 : d.nsg2;
    return d.g[comp - 1] + (size_t)(d.modular ? (s % ns) : s) * (size_t)d.cap + d.shift;
}
template <typename OT>
__device__ __forceinline__ GP<OT> hrowk(const Dir<OT> &d, int s, int comp) {
    if (comp == SR_C_M || !d.cold) return rowk(d, s, comp);
    return d.cold + ((size_t)(s % d.nsc) * 4 + (comp - 1)) * (size_t)d.cap + d.shift;
}

// ---- one score step, batched ------------------------------------------------
// Per-step uniform description of one aligner's rows (all unshifted: indexed
// by idx = k + shift >= 0).  Source levels that do not exist yet (score < 0)
// point at the aligner's NULL row, so the cell code has no branches on them.
template <typename OT>
struct StepRows {
    GP<OT> pMx, pMo1, pI1, pD1, pMo2, pI2, pD2;
    GP<OT> oM, oI1, oD1, oI2, oD2, cI1, cD1, cI2, cD2;
    int klo, khi, wlo, whi, k_end, s, shift, plen, tlen, begin, rev, pb, pe, tb, te;
};

template <typename OT, bool TWO>
__device__ __forceinline__ void step_rows(const Dir<OT> &d, const SrPen &pen, StepRows<OT> &r) {
    const int s = d.lvl + 1;
    const int R = reach(pen, s, d.begin);
    r.klo = max(-d.plen, -R); r.khi = min(d.tlen, R);
    r.wlo = max(-d.plen - 1, -R - pen.scope - 1); r.whi = min(d.tlen + 1, R + pen.scope + 1);
    r.k_end = d.tlen - d.plen; r.s = s; r.shift = d.shift; r.plen = d.plen; r.tlen = d.tlen;
    r.begin = d.begin; r.rev = d.rev; r.pb = d.pb; r.pe = d.pe; r.tb = d.tb; r.te = d.te;
    const GP<OT> nul = d.nullrow;
    r.pMx = (s >= pen.x) ? rowq(d, pen.x, SR_C_M) : nul;
    r.pMo1 = (s >= pen.o1 + pen.e1) ? rowq(d, pen.o1 + pen.e1, SR_C_M) : nul;
    r.pI1 = (s >= pen.e1) ? rowq(d, pen.e1, SR_C_I1) : nul;
    r.pD1 = (s >= pen.e1) ? rowq(d, pen.e1, SR_C_D1) : nul;
    r.pMo2 = r.pI2 = r.pD2 = nul;
    if (TWO) {
        if (s >= pen.o2 + pen.e2) r.pMo2 = rowq(d, pen.o2 + pen.e2, SR_C_M);
        if (s >= pen.e2) { r.pI2 = rowq(d, pen.e2, SR_C_I2); r.pD2 = rowq(d, pen.e2, SR_C_D2); }
    }
    r.oM = rowq(d, 0, SR_C_M); r.oI1 = rowq(d, 0, SR_C_I1); r.oD1 = rowq(d, 0, SR_C_D1);
    r.oI2 = rowq(d, 0, SR_C_I2); r.oD2 = rowq(d, 0, SR_C_D2);
    r.cI1 = r.cD1 = r.cI2 = r.cD2 = nullptr;
    if (d.cold) {
        r.cI1 = rowqc(d, SR_C_I1); r.cD1 = rowqc(d, SR_C_D1);
        r.cI2 = rowqc(d, SR_C_I2); r.cD2 = rowqc(d, SR_C_D2);
    }
}

// group g covers idx 4g .. 4g+3 (idx = k + shift)
template <typename OT, bool TWO>
__device__ __forceinline__ void group_load(const StepRows<OT> &r, int g, GroupIn<OT> &in) {
    const V4<OT> nv = {(OT)NULLV, (OT)NULLV, (OT)NULLV, (OT)NULLV};
    in.mo1 = in.i1 = in.d1 = in.mo2 = in.i2 = in.d2 = in.mx = nv;
    in.mo1L = in.mo1R = in.i1L = in.d1R = in.mo2L = in.mo2R = in.i2L = in.d2R = NULLV;
    const unsigned idx0 = (unsigned)g << 2;
    const int k0 = (int)idx0 - r.shift;
    if (r.s > 0 && k0 + 3 >= r.klo && k0 <= r.khi) {
        in.mo1 = ld4<OT>(r.pMo1, idx0); in.i1 = ld4<OT>(r.pI1, idx0); in.d1 = ld4<OT>(r.pD1, idx0);
        in.mx = ld4<OT>(r.pMx, idx0);
        if (TWO) { in.mo2 = ld4<OT>(r.pMo2, idx0); in.i2 = ld4<OT>(r.pI2, idx0); in.d2 = ld4<OT>(r.pD2, idx0); }
        if (k0 >= r.klo) {                     // left neighbour of cell 0 (idx0 - 1 >= 0 here)
            in.mo1L = (int)r.pMo1[idx0 - 1]; in.i1L = (int)r.pI1[idx0 - 1];
            if (TWO) { in.mo2L = (int)r.pMo2[idx0 - 1]; in.i2L = (int)r.pI2[idx0 - 1]; }
        }
        if (k0 + 3 <= r.khi) {                 // right neighbour of cell 3
            in.mo1R = (int)r.pMo1[idx0 + 4]; in.d1R = (int)r.pD1[idx0 + 4];
            if (TWO) { in.mo2R = (int)r.pMo2[idx0 + 4]; in.d2R = (int)r.pD2[idx0 + 4]; }
        }
    }
}

template <typename OT, bool TWO>
__device__ __forceinline__ void group_finish(const StepRows<OT> &r, int g, const GroupIn<OT> &in, LP P, LP T,
                                             int check_comp, int &my_ak, bool &my_reached) {
    const unsigned idx0 = (unsigned)g << 2;
    const int k0 = (int)idx0 - r.shift;
    int mv[4], i1v[4], i2v[4], d1v[4], d2v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = k0 + j;
        int m = NULLV, i1 = NULLV, i2 = NULLV, d1 = NULLV, d2 = NULLV;
        if (k >= r.klo && k <= r.khi) {
            if (r.s == 0) {
                if (k == 0) {
                    if (r.begin == SR_C_M) m = 0;
                    else if (r.begin == SR_C_I1) i1 = 0;
                    else if (r.begin == SR_C_I2) i2 = 0;
                    else if (r.begin == SR_C_D1) d1 = 0;
                    else d2 = 0;
                }
            } else {
                const unsigned lim = (unsigned)min(r.tlen, r.plen + k);
                const int a1 = (j == 0) ? in.mo1L : (int)in.mo1[j == 0 ? 0 : j - 1];
                const int b1 = (j == 0) ? in.i1L : (int)in.i1[j == 0 ? 0 : j - 1];
                const int c1 = (j == 3) ? in.mo1R : (int)in.mo1[j == 3 ? 3 : j + 1];
                const int e1 = (j == 3) ? in.d1R : (int)in.d1[j == 3 ? 3 : j + 1];
                i1 = bnd(max(a1, b1) + 1, lim);
                d1 = bnd(max(c1, e1), lim);
                if (TWO) {
                    const int a2 = (j == 0) ? in.mo2L : (int)in.mo2[j == 0 ? 0 : j - 1];
                    const int b2 = (j == 0) ? in.i2L : (int)in.i2[j == 0 ? 0 : j - 1];
                    const int c2 = (j == 3) ? in.mo2R : (int)in.mo2[j == 3 ? 3 : j + 1];
                    const int e2 = (j == 3) ? in.d2R : (int)in.d2[j == 3 ? 3 : j + 1];
                    i2 = bnd(max(a2, b2) + 1, lim);
                    d2 = bnd(max(c2, e2), lim);
                }
                m = bnd((int)in.mx[j] + 1, lim);
                m = max(m, max(max(i1, i2), max(d1, d2)));
                if (m < 0) m = NULLV;
            }
        }
        mv[j] = m; i1v[j] = i1; i2v[j] = i2; d1v[j] = d1; d2v[j] = d2;
    }
    // extension: the first 16-base window of all four cells is fetched from
    // LDS together; only cells whose whole window matched take the slow loop
    uint32_t xw[4];
    int nn[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        nn[j] = 0; xw[j] = 1u;
        if (mv[j] >= 0) {
            const int v = mv[j] - (k0 + j), h = mv[j];
            nn[j] = min(r.plen - v, r.tlen - h);
            if (nn[j] > 0) {
                if (!r.rev) xw[j] = win_fwd(P, r.pb + v) ^ win_fwd(T, r.tb + h);
                else xw[j] = win_rev(P, r.pe - 1 - v) ^ win_rev(T, r.te - 1 - h);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = k0 + j;
        if (mv[j] >= 0) {
            if (nn[j] > 0) {
                int c;
                if (!r.rev) c = xw[j] ? ((__ffs((int)xw[j]) - 1) >> SR_SYM_LOG) : SR_WIN;
                else c = xw[j] ? (__clz((int)xw[j]) >> SR_SYM_LOG) : SR_WIN;
                c = min(c, nn[j]);
                if (xw[j] == 0 && nn[j] > SR_WIN) {
                    const int v = mv[j] - k + SR_WIN, h = mv[j] + SR_WIN;
                    if (!r.rev) c += ext_fwd(P, T, r.pb + v, r.tb + h, nn[j] - SR_WIN);
                    else c += ext_rev(P, T, r.pe - 1 - v, r.te - 1 - h, nn[j] - SR_WIN);
                }
                mv[j] += c;
            }
            my_ak = max(my_ak, 2 * mv[j] - k);
        }
        if (k == r.k_end && k >= r.klo && k <= r.khi) {
            int val = mv[j];
            if (check_comp == SR_C_I1) val = i1v[j];
            else if (check_comp == SR_C_I2) val = i2v[j];
            else if (check_comp == SR_C_D1) val = d1v[j];
            else if (check_comp == SR_C_D2) val = d2v[j];
            if (val >= r.tlen) my_reached = true;
        }
    }
    V4<OT> oM, oI1, oI2, oD1, oD2;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        oM[j] = (OT)mv[j]; oI1[j] = (OT)i1v[j]; oI2[j] = (OT)i2v[j]; oD1[j] = (OT)d1v[j]; oD2[j] = (OT)d2v[j];
    }
    st4<OT>(r.oM, idx0, oM); st4<OT>(r.oI1, idx0, oI1); st4<OT>(r.oD1, idx0, oD1);
    if (TWO) { st4<OT>(r.oI2, idx0, oI2); st4<OT>(r.oD2, idx0, oD2); }
    if (r.cI1) {
        st4<OT>(r.cI1, idx0, oI1); st4<OT>(r.cD1, idx0, oD1);
        if (TWO) { st4<OT>(r.cI2, idx0, oI2); st4<OT>(r.cD2, idx0, oD2); }
    }
}

__device__ __forceinline__ void step_reduce(int ev, int side, int my_ak, bool my_reached) {
    my_ak = wave_max(my_ak);
    const int slot = ev % 3;
    if ((threadIdx.x & 63) == 0 && my_ak > 0) atomicMax(&g_sh.red_maxak[slot][side], my_ak);
    if (my_reached) g_sh.reached[slot][side] = 1;
    if (threadIdx.x == 0) { g_sh.red_maxak[(ev + 1) % 3][side] = 0; g_sh.reached[(ev + 1) % 3][side] = 0; }
}

// One score step: level lvl+1 of aligner dl[0] (when do_a) and of aligner
// dl[1] (when do_b) in the same pass.  When both run, even waves take aligner
// 0 and odd waves aligner 1 (the side is wave-uniform, so each wave sets up
// and walks ONE aligner); otherwise all waves share the one aligner.  A thread
// owns groups of 4 adjacent diagonals, two groups are in flight at a time (all
// source loads of both are issued before either is finished).  Reductions (M
// max antidiagonal, end test) go to LDS slot [ev % 3][side]; thread 0 advances
// the aligners' level counters.  The caller's __syncthreads() publishes it all.
template <typename OT, bool TWO, int NT>
__device__ __forceinline__ void wf_step_nl(int do_a_, int chk_a_, int do_b_, int ev_, int pen_sel_) {
    const int tid = threadIdx.x;
    const bool do_a = RFL(do_a_) != 0, do_b = RFL(do_b_) != 0;
    const int ev = RFL(ev_);
    const SrPen pen = load_pen(RFL(pen_sel_));
    const LP P = (LP)(lds_seq + RFL(g_sh.offP)), T = (LP)(lds_seq + RFL(g_sh.offT));
    constexpr int NW = NT / 64;
    const int wave = RFL(tid >> 6), lane = tid & 63;
    const bool both = do_a && do_b && NW >= 2;
    // which aligner this wave works on, its index among the waves of that side, and their number
    const int side = both ? (wave & 1) : (do_a ? 0 : 1);
    const int wis = both ? (wave >> 1) : wave;
    const int nws = both ? (NW >> 1) : NW;
    const int chk = (side == 0) ? RFL(chk_a_) : -1;
    if (do_a && do_b && NW < 2) {            // single-wave workgroup: the two aligners one after the other
        for (int sd = 0; sd < 2; sd++) {
            StepRows<OT> r;
            const Dir<OT> D = load_dir<OT>(sd, ev & 1);
            step_rows<OT, TWO>(D, pen, r);
            const int g0 = (r.wlo + r.shift) >> 2, g1 = (r.whi + r.shift) >> 2;
            int ak = 0; bool re = false;
            for (int g = g0 + lane; g <= g1; g += 64) {
                GroupIn<OT> in;
                group_load<OT, TWO>(r, g, in);
                group_finish<OT, TWO>(r, g, in, P, T, sd == 0 ? RFL(chk_a_) : -1, ak, re);
            }
            step_reduce(ev, sd, ak, re);
            if (tid == 0 && r.khi >= r.klo) g_sh.cells += (unsigned long long)(r.khi - r.klo + 1);
        }
    } else {
        StepRows<OT> r;
        const Dir<OT> D = load_dir<OT>(side, ev & 1);
        step_rows<OT, TWO>(D, pen, r);
        const int g0 = (r.wlo + r.shift) >> 2, g1 = (r.whi + r.shift) >> 2;
        const int stride = nws * 64;
        int ak = 0; bool re = false;
        for (int g = g0 + wis * 64 + lane; g <= g1; g += 2 * stride) {
            GroupIn<OT> i0, i1;
            const int gb = g + stride;
            group_load<OT, TWO>(r, g, i0);
            if (gb <= g1) group_load<OT, TWO>(r, gb, i1);
            group_finish<OT, TWO>(r, g, i0, P, T, chk, ak, re);
            if (gb <= g1) group_finish<OT, TWO>(r, gb, i1, P, T, chk, ak, re);
        }
        // every wave of a side reduces into that side's slot; waves of an idle side skip
        if ((side == 0 && do_a) || (side == 1 && do_b)) {
            ak = wave_max(ak);
            const int slot = ev % 3;
            if (lane == 0 && ak > 0) atomicMax(&g_sh.red_maxak[slot][side], ak);
            if (re) g_sh.reached[slot][side] = 1;
            if (lane == 0 && wis == 0 && r.khi >= r.klo)
                atomicAdd(&g_sh.cells, (unsigned long long)(r.khi - r.klo + 1));
        }
        if (tid == 0) {
            g_sh.red_maxak[(ev + 1) % 3][0] = 0; g_sh.reached[(ev + 1) % 3][0] = 0;
            g_sh.red_maxak[(ev + 1) % 3][1] = 0; g_sh.reached[(ev + 1) % 3][1] = 0;
        }
    }
    if (tid == 0) {
        dirl_advance(0, ev & 1, do_a);
        dirl_advance(1, ev & 1, do_b);
    }
}

__device__ __forceinline__ void bt_push(int op, int len, int &err) {
    if (len <= 0) return;
    int n = g_sh.bt_n;
    if (n > 0 && (int)(g_sh.bt_tmp[n - 1] & 15u) == op) { g_sh.bt_tmp[n - 1] += (uint32_t)len << 4; return; }
    if (n >= BT_TMP_CAP) { err |= SR_DEV_ERR_CIGAR_OVERFLOW; return; }
    g_sh.bt_tmp[n] = ((uint32_t)len << 4) | (uint32_t)op;
    g_sh.bt_n = n + 1;
}

// thread-0 backtrace over the full history of aligner dl[0] (oracle/wfa.c
// wfa_full); appends the segment's CIGAR to the pair's ops.
template <typename OT, bool TWO>
__device__ __noinline__ void backtrace_nl(int score, int cb, int ce, GP<uint32_t> ops, uint32_t cap) {
    const Dir<OT> d = load_dir<OT>(0);
    const SrPen pen = load_pen(0);
    int err = 0;
    const int plen = d.plen, tlen = d.tlen;
    int s = score, k = tlen - plen, comp = ce, off = tlen;
    g_sh.bt_n = 0;
    bool done = false;
    for (int guard = 0; guard < 4 * (plen + tlen) + 64 && !done; guard++) {
        if (comp == SR_C_M) {
            if (s == 0) {
                if (cb != SR_C_M || k != 0) err |= SR_DEV_ERR_BACKTRACE;
                bt_push(SR_OP_M, off, err);
                done = true; break;
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
            if (bty == 0 || bo > off) { err |= SR_DEV_ERR_BACKTRACE; break; }
            bt_push(SR_OP_M, off - bo, err);
            off = bo;
            switch (bty) {
            case 9: bt_push(SR_OP_X, 1, err); off -= 1; s -= pen.x; break;
            case 1: bt_push(SR_OP_I, 1, err); off -= 1; k -= 1; s -= pen.o1 + pen.e1; break;
            case 2: bt_push(SR_OP_I, 1, err); off -= 1; k -= 1; s -= pen.e1; comp = SR_C_I1; break;
            case 3: bt_push(SR_OP_I, 1, err); off -= 1; k -= 1; s -= pen.o2 + pen.e2; break;
            case 4: bt_push(SR_OP_I, 1, err); off -= 1; k -= 1; s -= pen.e2; comp = SR_C_I2; break;
            case 5: bt_push(SR_OP_D, 1, err); k += 1; s -= pen.o1 + pen.e1; break;
            case 6: bt_push(SR_OP_D, 1, err); k += 1; s -= pen.e1; comp = SR_C_D1; break;
            case 7: bt_push(SR_OP_D, 1, err); k += 1; s -= pen.o2 + pen.e2; break;
            default: bt_push(SR_OP_D, 1, err); k += 1; s -= pen.e2; comp = SR_C_D2; break;
            }
        } else {
            if (s == 0) {
                if (comp != cb || k != 0 || off != 0) err |= SR_DEV_ERR_BACKTRACE;
                done = true; break;
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
            else { err |= SR_DEV_ERR_BACKTRACE; break; }
            if ((take_ext ? c_ext : c_open) != off) { err |= SR_DEV_ERR_BACKTRACE; break; }
            if (is_ins) { bt_push(SR_OP_I, 1, err); off -= 1; k -= 1; }
            else { bt_push(SR_OP_D, 1, err); k += 1; }
            if (take_ext) s -= e; else { s -= o + e; comp = SR_C_M; }
        }
        if (s < 0) { err |= SR_DEV_ERR_BACKTRACE; break; }
    }
    if (!done) err |= SR_DEV_ERR_BACKTRACE;
    if (!err) {
        uint32_t cnt = g_sh.cig_cnt;
        for (int i = g_sh.bt_n - 1; i >= 0; i--)
            cig_append(ops, cnt, cap, (int)(g_sh.bt_tmp[i] & 15u), (int)(g_sh.bt_tmp[i] >> 4), err);
        g_sh.cig_cnt = cnt;
    }
    if (err) g_sh.err |= err;
}

// thread 0: describe the segment of aligner i
__device__ __forceinline__ void dirl_seg(DirL &l, const Seg &sg) {
    l.pb = sg.pb; l.pe = sg.pe; l.tb = sg.tb; l.te = sg.te;
    l.plen = sg.pe - sg.pb; l.tlen = sg.te - sg.tb;
}

// thread 0: ring layout of one direction:
// M[(scope_max+1)] | I1 I2 D1 D2 hot [ring_hot] each | cold [(scope_max+1)][4] | NULL row
template <typename OT>
__device__ __forceinline__ void dirl_ring(DirL &l, GP<OT> base, const SrAlignArgs &a, const SrPen &pen, bool with_cold) {
    const size_t cap = (size_t)a.ring_cap;
    l.m = (unsigned long long)base;
    GP<OT> p = base + (size_t)(a.ring_scope + 1) * cap;
    for (int c = 0; c < 4; c++) { l.g[c] = (unsigned long long)p; p += (size_t)a.ring_hot * cap; }
    l.cold = with_cold ? (unsigned long long)p : 0ULL;
    l.nullrow = (unsigned long long)(p + (size_t)(a.ring_scope + 1) * 4 * cap);
    l.nsm = pen.scope + 1; l.nsg1 = pen.e1 + 2; l.nsg2 = pen.two ? pen.e2 + 2 : 2; l.nsc = pen.scope + 1;
    l.cap = a.ring_cap; l.modular = 1;
    dirl_reset(l);
}

// plain WFA with full history + backtrace on one segment (all threads)
template <typename OT, bool TWO, int NT>
__device__ __forceinline__ void wfa_base(const Seg &sg, const SrAlignArgs &a, GP<OT> hist, int &ev,
                                         unsigned long long &steps, GP<uint32_t> ops, uint32_t cap) {
    if (threadIdx.x == 0) {
        DirL &l = g_sh.dl[0];
        const size_t comp_stride = (size_t)a.hist_levels * (size_t)a.hist_w;
        l.m = (unsigned long long)hist;
        for (int c = 0; c < 4; c++) l.g[c] = (unsigned long long)(hist + (size_t)(c + 1) * comp_stride);
        l.cold = 0ULL; l.nsm = l.nsg1 = l.nsg2 = l.nsc = a.hist_levels;
        l.nullrow = (unsigned long long)(hist + 5 * comp_stride);
        l.cap = a.hist_w; l.shift = a.hist_w / 2; l.modular = 0; l.rev = 0; l.begin = sg.cb;
        dirl_seg(l, sg);
        dirl_reset(l);
    }
    __syncthreads();
    int s = 0;
    bool found = false;
    for (;;) {
        wf_step_nl<OT, TWO, NT>(1, sg.ce, 0, ev, 0);
        __syncthreads();
        const bool r = RFL(g_sh.reached[ev % 3][0]) != 0;
        ev++;
        steps++;
        if (r) { found = true; break; }
        if (s + 1 >= a.hist_levels) break;
        s++;
    }
    if (threadIdx.x == 0) {
        if (!found) g_sh.err |= SR_DEV_ERR_BASE_OVERFLOW;
        else backtrace_nl<OT, TWO>(s, sg.cb, sg.ce, ops, cap);
    }
    __syncthreads();
}

template <typename OT>
__device__ __forceinline__ void bp_try(const Dir<OT> &d0, const Dir<OT> &d1, int c, int gap,
                                       int score_0, int score_i, int kinv, bool bp_forward) {
    const int k0 = g_sh.cand[c];
    if (k0 == INT_MAX) return;
    if (!(score_0 + score_i - gap < g_sh.bp_score)) return;
    const int k1 = kinv - k0;
    const int o0 = (int)hrowk(d0, score_0, c)[k0];
    const int o1 = (int)hrowk(d1, score_i, c)[k1];
    if (bp_forward) {
        g_sh.bp_score_f = score_0; g_sh.bp_score_r = score_i;
        g_sh.bp_k_f = k0; g_sh.bp_k_r = k1; g_sh.bp_off_f = o0; g_sh.bp_off_r = o1;
    } else {
        g_sh.bp_score_f = score_i; g_sh.bp_score_r = score_0;
        g_sh.bp_k_f = k1; g_sh.bp_k_r = k0; g_sh.bp_off_f = o1; g_sh.bp_off_r = o0;
    }
    g_sh.bp_score = score_0 + score_i - gap;
    g_sh.bp_comp = c;
}

// breakpoint detection between level score_0 of aligner a0 and the last
// `scope` levels of the other aligner (oracle/wfa.c bialign_overlap)
template <typename OT, bool TWO, int NT>
__device__ __noinline__ void bi_overlap_nl(int a0_, int score_0_, int score_1_, int bp_forward_) {
    const int tid = threadIdx.x;
    const int a0 = RFL(a0_), a1 = 1 - a0, score_0 = RFL(score_0_), score_1 = RFL(score_1_);
    const bool bp_forward = RFL(bp_forward_) != 0;
    const Dir<OT> d0 = load_dir<OT>(a0), d1 = load_dir<OT>(a1);
    const SrPen pen = load_pen(0);
    const int plen = d0.plen, tlen = d0.tlen;
    const int kinv = tlen - plen;
    const int mak0 = RFL(g_sh.mak[a0][score_0 % d0.nsm]);
    const int gapmax = TWO ? max(pen.o1, pen.o2) : pen.o1;
    const int R0 = reach(pen, score_0, d0.begin);
    const int klo0 = max(-plen, -R0), khi0 = min(tlen, R0);
    for (int i = 0; i < pen.scope; i++) {
        const int score_i = score_1 - i;
        if (score_i < 0) break;
        const int mak1 = RFL(g_sh.mak[a1][score_i % d1.nsm]);
        if (mak0 + mak1 < plen + tlen) continue;            // no diagonal can overlap
        if (score_0 + score_i - gapmax >= RFL(g_sh.bp_score)) continue;
        if (tid < 5) g_sh.cand[tid] = INT_MAX;
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
                if (o0 >= 0 && o1 >= 0 && o0 + o1 >= tlen) atomicMin(&g_sh.cand[c], k0);
            }
        }
        __syncthreads();
        if (tid == 0) {
            // order: D2, I2, D1, I1, M with the running thresholds
            if (TWO) {
                bp_try(d0, d1, SR_C_D2, pen.o2, score_0, score_i, kinv, bp_forward);
                bp_try(d0, d1, SR_C_I2, pen.o2, score_0, score_i, kinv, bp_forward);
            }
            bp_try(d0, d1, SR_C_D1, pen.o1, score_0, score_i, kinv, bp_forward);
            bp_try(d0, d1, SR_C_I1, pen.o1, score_0, score_i, kinv, bp_forward);
            bp_try(d0, d1, SR_C_M, 0, score_0, score_i, kinv, bp_forward);
        }
        __syncthreads();
    }
}

// computes F level done_f+1 and/or R level done_r+1 in one pass (one barrier)
#define STEP_BOTH(DO_F, DO_R)                                                                      \
    do {                                                                                           \
        const bool do_f_ = (DO_F), do_r_ = (DO_R);                                                 \
        wf_step_nl<OT, TWO, NT>(do_f_, -1, do_r_, ev, 0);                                          \
        __syncthreads();                                                                           \
        if (do_f_) {                                                                               \
            done_f++; ak_f_done = RFL(g_sh.red_maxak[ev % 3][0]);                                  \
            if (tid == 0) g_sh.mak[0][mslot_f] = ak_f_done;                                        \
            mslot_f = slot_inc(mslot_f, nsm); steps++;                                             \
        }                                                                                          \
        if (do_r_) {                                                                               \
            done_r++; ak_r_done = RFL(g_sh.red_maxak[ev % 3][1]);                                  \
            if (tid == 0) g_sh.mak[1][mslot_r] = ak_r_done;                                        \
            mslot_r = slot_inc(mslot_r, nsm); steps++;                                             \
        }                                                                                          \
        ev++;                                                                                      \
    } while (0)

// biWFA breakpoint search on one segment (oracle/wfa.c bialign_find_breakpoint).
// Forward and reverse aligners are stepped together: the level the other
// aligner will need next is computed speculatively in the same pass (it does
// not depend on this one), so one barrier serves two wavefront steps.
template <typename OT, bool TWO, int NT>
__device__ __forceinline__ bool find_breakpoint(const Seg &sg, const SrAlignArgs &a, GP<OT> ring, int &ev,
                                                unsigned long long &steps) {
    const int tid = threadIdx.x;
    const SrPen pen = a.pen;
    if (tid == 0) {
        DirL &F = g_sh.dl[0], &R = g_sh.dl[1];
        dirl_ring<OT>(F, ring, a, pen, true);
        dirl_ring<OT>(R, ring + a.ring_dir_stride, a, pen, true);
        dirl_seg(F, sg); dirl_seg(R, sg);
        F.rev = 0; R.rev = 1; F.begin = sg.cb; R.begin = sg.ce;
        F.shift = F.plen + 1; R.shift = R.plen + 1;
        g_sh.bp_score = INT_MAX;
    }
    __syncthreads();
    const int plen = sg.pe - sg.pb, tlen = sg.te - sg.tb;
    const int max_antidiagonal = plen + tlen - 1;
    const int scope = pen.scope, nsm = pen.scope + 1;
    const int gap_opening = TWO ? max(pen.o1, pen.o2) : pen.o1;
    const long long smax = 2LL * ((long long)pen.o1 * 2 + (long long)pen.e1 * (plen + tlen)) + 1024;
    int score_f = 0, score_r = 0;
    int done_f = -1, done_r = -1;        // highest level computed
    int mslot_f = 0, mslot_r = 0;        // (done + 1) % nsm
    int ak_f_done = 0, ak_r_done = 0;    // max antidiagonal of level done_f / done_r
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
            if (score_f + min_score_r - gap_opening >= RFL(g_sh.bp_score)) break;
            bi_overlap_nl<OT, TWO, NT>(0, score_f, score_r, 1);
            ++score_r;
            if (done_r < score_r) STEP_BOTH(done_f < score_f + 1, true);
        }
        const int min_score_f = (score_f > scope - 1) ? score_f - (scope - 1) : 0;
        __syncthreads();
        if (min_score_f + score_r - gap_opening >= RFL(g_sh.bp_score)) break;
        bi_overlap_nl<OT, TWO, NT>(1, score_r, score_f, 0);
        ++score_f;
        if (done_f < score_f) STEP_BOTH(true, done_r < score_r + 1);
        last_wf_forward = true;
        if ((long long)score_f + score_r > smax) { ok = false; break; }
    }
    __syncthreads();
    if (ok && RFL(g_sh.bp_score) == INT_MAX) ok = false;
    if (!ok && tid == 0) g_sh.err |= SR_DEV_ERR_SCORE_BOUND;
    __syncthreads();
    return ok;
}

// score-only end-to-end WFA (orientation check); returns INT_MAX when the
// score exceeds max_score (max_score < 0: unbounded)
template <typename OT, int NT>
__device__ __forceinline__ int wfa_score_only(int plen, int tlen, const SrAlignArgs &a, GP<OT> ring, int max_score,
                                              int &ev, unsigned long long &steps) {
    const SrPen pen = a.ori;
    if (threadIdx.x == 0) {
        DirL &l = g_sh.dl[0];
        dirl_ring<OT>(l, ring, a, pen, false);       // no breakpoint detection here: no history
        l.rev = 0; l.pb = 0; l.pe = plen; l.tb = 0; l.te = tlen; l.plen = plen; l.tlen = tlen;
        l.shift = plen + 1; l.begin = SR_C_M;
    }
    __syncthreads();
    const long long smax = (long long)pen.o1 * 2 + (long long)pen.e1 * (plen + tlen) + 64;
    int s = 0, res = INT_MAX;
    for (;;) {
        wf_step_nl<OT, false, NT>(1, SR_C_M, 0, ev, 1);
        __syncthreads();
        const bool r = RFL(g_sh.reached[ev % 3][0]) != 0;
        ev++; steps++;
        if (r) { res = s; break; }
        if (s > smax) { if (threadIdx.x == 0) g_sh.err |= SR_DEV_ERR_SCORE_BOUND; break; }
        if (max_score >= 0 && s >= max_score) break;
        s++;
    }
    __syncthreads();
    return res;
}

template <typename OT, int NT, bool TWO>
__global__ void __launch_bounds__(NT, SR_MIN_WAVES) sr_align_kernel(SrAlignArgs a) {
    const int tid = threadIdx.x;
    GP<OT> ring = (GP<OT>)(OT *)a.ring + (size_t)blockIdx.x * a.ring_wg_stride;
    GP<OT> hist = (GP<OT>)(OT *)a.hist + (size_t)blockIdx.x * a.hist_wg_stride;
    unsigned long long steps = 0, nbase = 0, nbp = 0;
    unsigned long long t_ori = 0, t_bp = 0, t_base = 0, t_bp_top = 0, t_all = 0;   // 100 MHz ticks (thread 0's view)
    int ev = 0;
    {   // NULL rows: one per ring direction and one for the base-case history
        const size_t cap = (size_t)a.ring_cap;
        GP<OT> n0 = ring + a.ring_dir_stride - cap, n1 = ring + 2 * a.ring_dir_stride - cap;
        for (int i = tid; i < a.ring_cap; i += NT) { n0[i] = (OT)NULLV; n1[i] = (OT)NULLV; }
        GP<OT> nh = hist + (size_t)5 * a.hist_levels * a.hist_w;
        for (int i = tid; i < a.hist_w; i += NT) nh[i] = (OT)NULLV;
        if (tid == 0) { g_sh.pen[0] = a.pen; g_sh.pen[1] = a.ori; g_sh.cells = 0; }
        __syncthreads();
    }
    for (;;) {
        if (tid == 0) {
            g_sh.pair = (int)atomicAdd(a.queue_head, 1u);
            g_sh.err = 0; g_sh.sp = 0; g_sh.cig_cnt = 0; g_sh.score_acc = 0;
            for (int i = 0; i < 3; i++) { g_sh.red_maxak[i][0] = g_sh.red_maxak[i][1] = 0; g_sh.reached[i][0] = g_sh.reached[i][1] = 0; }
        }
        __syncthreads();
        int pair = RFL(g_sh.pair);
        if (pair >= (int)a.npairs) break;
        if (a.order) pair = (int)a.order[pair];
        const uint32_t q = a.pair_q[pair], t = a.pair_t[pair];
        const int plen = (int)a.seqlen[q], tlen = (int)a.seqlen[t];
        const int pw = SR_SEQ_WORDS(plen), tw = SR_SEQ_WORDS(tlen);
        uint32_t *Pf = lds_seq, *Pr = lds_seq + a.max_words, *Tt = lds_seq + 2 * (size_t)a.max_words;
        load_seq_lds<NT>(Pf, (GP<const uint32_t>)a.seqwords + a.word_off_fwd[q] - 1, pw);
        load_seq_lds<NT>(Pr, (GP<const uint32_t>)a.seqwords + a.word_off_rc[q] - 1, pw);
        load_seq_lds<NT>(Tt, (GP<const uint32_t>)a.seqwords + a.word_off_fwd[t] - 1, tw);
        if (tid == 0) { g_sh.offP = 1; g_sh.offT = 2 * (int)a.max_words + 1; }
        __syncthreads();
        const unsigned long long tk0 = __builtin_amdgcn_s_memrealtime();
        // ---- orientation (forward on ties; reverse scored only up to fwd-1)
        const int fwd = wfa_score_only<OT, NT>(plen, tlen, a, ring, -1, ev, steps);
        int rev = INT_MAX;
        bool is_rev = false;
        if (fwd > 0 && fwd != INT_MAX) {
            if (tid == 0) g_sh.offP = (int)a.max_words + 1;
            __syncthreads();
            rev = wfa_score_only<OT, NT>(plen, tlen, a, ring, fwd - 1, ev, steps);
            is_rev = rev < fwd;
        }
        if (tid == 0) g_sh.offP = is_rev ? (int)a.max_words + 1 : 1;
        __syncthreads();
        const unsigned long long tk1 = __builtin_amdgcn_s_memrealtime();
        t_ori += tk1 - tk0;
        GP<uint32_t> ops = (GP<uint32_t>)a.cigar_ops + a.cigar_base[pair];
        const uint32_t cap = (uint32_t)(a.cigar_base[pair + 1] - a.cigar_base[pair]);
        // ---- main alignment
        if (tid == 0) {
            Seg s0; s0.pb = 0; s0.pe = plen; s0.tb = 0; s0.te = tlen; s0.cb = SR_C_M; s0.ce = SR_C_M;
            s0.score_rem = (a.mem_mode == 3) ? INT_MAX : -1;   // -1: force plain WFA
            g_sh.stack[0] = s0; g_sh.sp = 1;
        }
        __syncthreads();
        while (RFL(g_sh.sp) > 0 && RFL(g_sh.err) == 0) {
            const int top = RFL(g_sh.sp) - 1;
            Seg sg;
            sg.pb = RFL(g_sh.stack[top].pb); sg.pe = RFL(g_sh.stack[top].pe);
            sg.tb = RFL(g_sh.stack[top].tb); sg.te = RFL(g_sh.stack[top].te);
            sg.cb = RFL(g_sh.stack[top].cb); sg.ce = RFL(g_sh.stack[top].ce);
            sg.score_rem = RFL(g_sh.stack[top].score_rem);
            __syncthreads();
            if (tid == 0) g_sh.sp = top;
            const int sp_len = sg.pe - sg.pb, st_len = sg.te - sg.tb;
            if (st_len == 0 || sp_len == 0) {
                if (tid == 0) {
                    int err = 0; uint32_t cnt = g_sh.cig_cnt;
                    if (st_len == 0) cig_append(ops, cnt, cap, SR_OP_D, sp_len, err);
                    else cig_append(ops, cnt, cap, SR_OP_I, st_len, err);
                    g_sh.cig_cnt = cnt; if (err) g_sh.err |= err;
                }
                __syncthreads();
                continue;
            }
            const bool base = sg.score_rem <= 250 || max(sp_len, st_len) <= 100;
            if (base) {
                nbase++;
                const unsigned long long tb0 = __builtin_amdgcn_s_memrealtime();
                wfa_base<OT, TWO, NT>(sg, a, hist, ev, steps, ops, cap);
                t_base += __builtin_amdgcn_s_memrealtime() - tb0;
                continue;
            }
            nbp++;
            const unsigned long long tp0 = __builtin_amdgcn_s_memrealtime();
            const bool ok = find_breakpoint<OT, TWO, NT>(sg, a, ring, ev, steps);
            {
                const unsigned long long dtp = __builtin_amdgcn_s_memrealtime() - tp0;
                t_bp += dtp;
                if (sg.score_rem == INT_MAX) t_bp_top += dtp;
            }
            if (ok && tid == 0) {
                const int bh = g_sh.bp_off_f, bv = g_sh.bp_off_f - g_sh.bp_k_f;
                if (bv < 0 || bv > sp_len || bh < 0 || bh > st_len) g_sh.err |= SR_DEV_ERR_BREAKPOINT;
                else if (g_sh.sp + 2 > SR_STACK_DEPTH) g_sh.err |= SR_DEV_ERR_STACK;
                else {
                    Seg h1; h1.pb = sg.pb + bv; h1.pe = sg.pe; h1.tb = sg.tb + bh; h1.te = sg.te;
                    h1.cb = g_sh.bp_comp; h1.ce = sg.ce; h1.score_rem = g_sh.bp_score_r;
                    Seg h0; h0.pb = sg.pb; h0.pe = sg.pb + bv; h0.tb = sg.tb; h0.te = sg.tb + bh;
                    h0.cb = sg.cb; h0.ce = g_sh.bp_comp; h0.score_rem = g_sh.bp_score_f;
                    g_sh.stack[g_sh.sp] = h1; g_sh.stack[g_sh.sp + 1] = h0; g_sh.sp += 2;
                }
            }
            __syncthreads();
        }
        __syncthreads();
        // ---- score of the final CIGAR (gaps costed once per merged run)
        const uint32_t cnt = RFL(g_sh.cig_cnt);
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
        if (part) atomicAdd(&g_sh.score_acc, part);
        __syncthreads();
        if (tid == 0) {
            a.is_reverse[pair] = is_rev ? 1 : 0;
            a.score[pair] = g_sh.err ? -1 : g_sh.score_acc;
            a.ori_fwd[pair] = fwd; a.ori_rev[pair] = rev;
            a.cigar_cnt[pair] = g_sh.err ? 0u : cnt;
            if (g_sh.err) atomicOr(a.error_flag, g_sh.err);
        }
        __syncthreads();
        t_all += __builtin_amdgcn_s_memrealtime() - tk0;
    }
    if (tid == 0) {
        atomicAdd(&a.counters[0], g_sh.cells);
        atomicAdd(&a.counters[1], steps);
        atomicAdd(&a.counters[2], nbase);
        atomicAdd(&a.counters[3], nbp);
        atomicAdd(&a.counters[6], t_ori);
        atomicAdd(&a.counters[7], t_bp);
        atomicAdd(&a.counters[8], t_base);
        atomicAdd(&a.counters[9], t_bp_top);
        atomicAdd(&a.counters[10], t_all);
    }
}

// ------------------------------------------------------------------ launcher
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
extern "C" int srk_align_v3(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (off16) {
        if (nthreads == 64) return launch_align<int16_t, 64>(a, nwg, lds_bytes, st);
        if (nthreads == 128) return launch_align<int16_t, 128>(a, nwg, lds_bytes, st);
        return launch_align<int16_t, 256>(a, nwg, lds_bytes, st);
    }
    if (nthreads == 64) return launch_align<int32_t, 64>(a, nwg, lds_bytes, st);
    if (nthreads == 128) return launch_align<int32_t, 128>(a, nwg, lds_bytes, st);
    return launch_align<int32_t, 256>(a, nwg, lds_bytes, st);
}
