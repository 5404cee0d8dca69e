// sr_orient.hip -- orientation pass as its own kernel: one pair per WAVE, no workgroup barriers.
//
// Rule (the allwave source is absent): the query is scored
// forward and reverse-complemented against the target with the one-piece orientation penalties
// (--orientation-scores, default 0,1,1,1); both aligners advance in lockstep, one score level at a time, and
// the first to reach the end decides (forward on ties) -- the same predicate as "reverse iff strictly lower".
//
// With mismatch 1 there is nothing to block over (M[s] needs M[s-1]), so a level is little work (2 aligners x
// 1..5 wave tiles); run by a 4-wave workgroup it costs two barriers and a serial set-up per level.  Here every
// wave owns a pair: levels follow each other without barriers (a wave sees its own stores in order), the
// ranges are scalar arithmetic, and 16 pairs per CU keep the SIMDs busy.  Tile code = sr_align_blk.inc's
// blk_tile for one level (4 diagonals per lane, DPP neighbours, 2 halo lanes).
#include "sr_dev_common.h"
#include <cstdlib>
namespace SR_NS {

__device__ __forceinline__ int o_lane_left(int x) { return __builtin_amdgcn_update_dpp(NULLV, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int o_lane_right(int x) { return __builtin_amdgcn_update_dpp(NULLV, x, 0x130, 0xf, 0xf, false); }

template <typename OT>
__global__ void __launch_bounds__(64, 4) sr_orient_kernel(SrAlignArgs a) {
    const int lane = threadIdx.x;
    const SrPen pen = a.ori;
    const unsigned w = (unsigned)a.orow;
    const int depth = pen.scope + 1;
    GP<OT> ring = (GP<OT>)(OT *)a.oring + (size_t)blockIdx.x * a.oring_wg_stride;
    GP<OT> nul = ring + (size_t)depth * 3 * w;
    for (unsigned i = lane; i < w; i += 64) nul[i] = (OT)NULLV;
    unsigned long long cells = 0, steps = 0;
    int err = 0;
#define OROW(LVL, C) (((LVL) < 0) ? nul : ring + (size_t)(((unsigned)((LVL) % depth) * 3u + (unsigned)(C)) * w))
    for (;;) {
        int pair = 0;
        if (lane == 0) pair = (int)atomicAdd(a.oqueue, 1u);
        pair = RFL(pair);
        if (pair >= (int)a.npairs) break;
        if (a.order) pair = (int)a.order[pair];                 // cost-sorted dequeue order
        const uint32_t q = a.pair_q[pair], t = a.pair_t[pair];
        const int plen = (int)a.seqlen[q], tlen = (int)a.seqlen[t];
        const int pw = SR_SEQ_WORDS(plen), tw = SR_SEQ_WORDS(tlen);
        __syncthreads();                                        // (one wave: orders the LDS reuse)
        load_seq_lds<64>(lds_seq, (GP<const uint32_t>)a.seqwords + a.word_off_fwd[q] - 1, pw);
        load_seq_lds<64>(lds_seq + a.max_words, (GP<const uint32_t>)a.seqwords + a.word_off_rc[q] - 1, pw);
        load_seq_lds<64>(lds_seq + 2 * (size_t)a.max_words, (GP<const uint32_t>)a.seqwords + a.word_off_fwd[t] - 1, tw);
        __syncthreads();
        const LP T = (LP)(lds_seq + 2 * (size_t)a.max_words + 1);
        const int shift = plen + 9, width = (plen + tlen + 32) & ~3, kend = tlen - plen;
        int fwd = -1, rev = INT_MAX, is_rev = 0;
        const long long smax = (long long)pen.o1 * 2 + (long long)pen.e1 * (plen + tlen) + 64;
        for (int s = 0;; s++) {
            const int R = reach(pen, s, SR_C_M);
            const int klo = max(-plen, -R), khi = min(tlen, R);
            const int wlo = max(-plen - 1, -R - pen.scope - 1), whi = min(tlen + 1, R + pen.scope + 1);
            const int glo = (wlo + shift) >> 2, ghi = (whi + shift) >> 2;
            const int nt = (ghi - glo + 60) / 60;
            cells += 2ull * (unsigned long long)(khi - klo + 1);
            steps += 2;
            GP<OT> rMx = OROW(s - pen.x, 0), rMo = OROW(s - pen.o1 - pen.e1, 0), rI = OROW(s - pen.e1, 1), rD = OROW(s - pen.e1, 2);
            GP<OT> oM = OROW(s, 0), oI = OROW(s, 1), oD = OROW(s, 2);
            bool hit[2] = {false, false};
#pragma unroll
            for (int job = 0; job < 2; job++) {
                const LP P = (LP)(lds_seq + (job ? a.max_words : 0) + 1);
                const int base = job * width;
                for (int ti = 0; ti < nt; ti++) {
                    const int g = glo + ti * 60 + lane - 2;
                    const bool owned = (lane >= 2) && (lane < 62) && (g <= ghi);
                    const int k0 = (g << 2) - shift;
                    const unsigned idx0 = (unsigned)(base + (g << 2));
                    const V4<OT> vmx = ld4<OT>(rMx, idx0), vmo = ld4<OT>(rMo, idx0), vi = ld4<OT>(rI, idx0), vd = ld4<OT>(rD, idx0);
                    int mx[4], mo[4], si[4], sd[4];
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) { mx[qq] = (int)vmx[qq]; mo[qq] = (int)vmo[qq]; si[qq] = (int)vi[qq]; sd[qq] = (int)vd[qq]; }
                    const int moL = o_lane_left(mo[3]), moR = o_lane_right(mo[0]);
                    const int iL = o_lane_left(si[3]), dR = o_lane_right(sd[0]);
                    int mv[4], i1v[4], d1v[4];
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const int k = k0 + qq;
                        const bool inr = (k >= klo) && (k <= khi);
                        const unsigned lim = (unsigned)min(tlen, plen + k);
                        const int a1 = (qq == 0) ? moL : mo[qq == 0 ? 0 : qq - 1];
                        const int b1 = (qq == 0) ? iL : si[qq == 0 ? 0 : qq - 1];
                        const int c1 = (qq == 3) ? moR : mo[qq == 3 ? 3 : qq + 1];
                        const int f1 = (qq == 3) ? dR : sd[qq == 3 ? 3 : qq + 1];
                        int i1 = bnd(max(a1, b1) + 1, lim);
                        int d1 = bnd(max(c1, f1), lim);
                        int m = bnd(mx[qq] + 1, lim);
                        m = max(m, max(i1, d1));
                        if (!inr) { m = NULLV; i1 = NULLV; d1 = NULLV; }
                        if (s == 0) { m = (inr && k == 0) ? 0 : NULLV; i1 = NULLV; d1 = NULLV; }
                        mv[qq] = m; i1v[qq] = i1; d1v[qq] = d1;
                    }
                    // extension (same scheme as blk_tile: branch-free first window, then only the cell positions
                    // some lane still extends)
                    int more = 0;
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const bool valid = owned && mv[qq] >= 0;
                        const int h = valid ? mv[qq] : 0, v = valid ? mv[qq] - (k0 + qq) : 0;
                        const int nn = valid ? min(plen - v, tlen - h) : 0;
                        const uint32_t xw = win_fwd(P, v) ^ win_fwd(T, h);
                        const unsigned z = (unsigned)(__ffs((int)xw) - 1) >> SR_SYM_LOG;
                        const int c = (int)min(min(z, (unsigned)SR_WIN), (unsigned)nn);
                        mv[qq] += c;
                        more |= (xw == 0u && nn > SR_WIN) ? (1 << qq) : 0;
                    }
                    unsigned long long pend[4];
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) pend[qq] = __ballot((more >> qq) & 1);
                    while ((pend[0] | pend[1] | pend[2] | pend[3]) != 0ull) {
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) {
                            if (pend[qq] == 0ull) continue;
                            const bool on = (more >> qq) & 1;
                            const int h = on ? mv[qq] : 0, v = on ? mv[qq] - (k0 + qq) : 0;
                            const int nn = on ? min(plen - v, tlen - h) : 0;
                            const uint32_t xw = win_fwd(P, v) ^ win_fwd(T, h);
                            const unsigned z = (unsigned)(__ffs((int)xw) - 1) >> SR_SYM_LOG;
                            const int c = (int)min(min(z, (unsigned)SR_WIN), (unsigned)nn);
                            mv[qq] += c;
                            if (!(xw == 0u && nn > SR_WIN)) more &= ~(1 << qq);
                            pend[qq] = __ballot((more >> qq) & 1);
                        }
                    }
#pragma unroll
                    for (int qq = 0; qq < 4; qq++)
                        hit[job] |= owned && (k0 + qq) == kend && (k0 + qq) >= klo && (k0 + qq) <= khi && mv[qq] >= tlen;
                    if (owned) {
                        V4<OT> vM, vI, vD;
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) { vM[qq] = (OT)mv[qq]; vI[qq] = (OT)i1v[qq]; vD[qq] = (OT)d1v[qq]; }
                        st4<OT>(oM, idx0, vM); st4<OT>(oI, idx0, vI); st4<OT>(oD, idx0, vD);
                    }
                }
            }
            __syncthreads();                                    // one wave: workgroup-scope fence, level s is visible
            const bool rf = __ballot(hit[0]) != 0ull, rr = __ballot(hit[1]) != 0ull;
            if (rf) { fwd = s; break; }
            if (rr) { rev = s; is_rev = 1; break; }
            if (s > smax) { err |= SR_DEV_ERR_SCORE_BOUND; break; }
        }
        if (lane == 0) {
            a.is_reverse[pair] = is_rev ? 1 : 0;
            a.ori_fwd[pair] = fwd; a.ori_rev[pair] = rev;
        }
    }
#undef OROW
    if (lane == 0) {
        if (cells) { atomicAdd(&a.counters[0], cells); atomicAdd(&a.counters[6], cells); }
        if (steps) atomicAdd(&a.counters[1], steps);
        if (err) atomicOr(a.error_flag, err);
    }
}

// ---- blocked instance for the default orientation penalties (mismatch 1, gap-open 1, gap-extend 1, one piece) -----
// With x = 1, o + e = 2, e = 1 every cell of level s depends on levels s-1 and s-2 at diagonals k-1 .. k+1 only: a
// wave tile keeps M[s-1], M[s-2], I[s-1], D[s-1] of its 4 diagonals per lane in registers and walks OB levels without
// touching memory -- neighbours by DPP, halo of one diagonal per level and side (2 lanes for OB = 8).  Rows are read
// and written once per OB levels (4 + 4 per tile) instead of 4 + 3 per level, and the store -> load round trip between
// consecutive levels, which is what the level-by-level kernel spends its time on, is paid once per block.
// Two row sets (block parity) of {M last, M last-1, I last, D last}; a block's window = the last level's range + OB + 2.
#define OB 8
template <typename OT>
__global__ void __launch_bounds__(64, 4) sr_orient_blk_kernel(SrAlignArgs a) {
    const int lane = threadIdx.x;
    const SrPen pen = a.ori;
    const unsigned w = (unsigned)a.orow;
    GP<OT> ring = (GP<OT>)(OT *)a.oring + (size_t)blockIdx.x * a.oring_wg_stride;
    GP<OT> nul = ring + (size_t)8 * w;
    for (unsigned i = lane; i < w; i += 64) nul[i] = (OT)NULLV;
    unsigned long long cells = 0, steps = 0;
    int err = 0;
    for (;;) {
        int pair = 0;
        if (lane == 0) pair = (int)atomicAdd(a.oqueue, 1u);
        pair = RFL(pair);
        if (pair >= (int)a.npairs) break;
        if (a.order) pair = (int)a.order[pair];
        const uint32_t q = a.pair_q[pair], t = a.pair_t[pair];
        const int plen = (int)a.seqlen[q], tlen = (int)a.seqlen[t];
        const int pw = SR_SEQ_WORDS(plen), tw = SR_SEQ_WORDS(tlen);
        __syncthreads();
        load_seq_lds<64>(lds_seq, (GP<const uint32_t>)a.seqwords + a.word_off_fwd[q] - 1, pw);
        load_seq_lds<64>(lds_seq + a.max_words, (GP<const uint32_t>)a.seqwords + a.word_off_rc[q] - 1, pw);
        load_seq_lds<64>(lds_seq + 2 * (size_t)a.max_words, (GP<const uint32_t>)a.seqwords + a.word_off_fwd[t] - 1, tw);
        __syncthreads();
        const LP T = (LP)(lds_seq + 2 * (size_t)a.max_words + 1);
        const int shift = plen + 24, width = (plen + tlen + 64) & ~3, kend = tlen - plen;
        int fwd = -1, rev = INT_MAX, is_rev = 0;
        const long long smax = (long long)pen.o1 * 2 + (long long)pen.e1 * (plen + tlen) + 64;
        // Lower bound of the reverse-complement orientation's score from the 8-mers its query shares with the target
        // (a.kbits, see sr_kmer_bits_kernel): below that level the reverse aligner cannot reach the end, so only the forward
        // one runs; if the forward aligner has not finished by then, the reverse one catches up block by block (its rows
        // live in columns of their own) and the two go on in lockstep.  Outcome and scores are those of the lockstep.
        int lb_rev = 0;
#if SR_SYMBITS == 2
        if (a.kbits && plen >= 8) {
            const uint32_t *bm = a.kbits + (size_t)t * 2048;
            const LP Prc = (LP)(lds_seq + a.max_words + 1);
            const int npos = plen - 7;
            int cnt = 0;
            for (int i = lane; i < npos; i += 64) {
                const uint32_t code = win_fwd(Prc, i) & 0xffffu;
                cnt += (int)((bm[code >> 5] >> (code & 31u)) & 1u);
            }
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            lb_rev = (npos - cnt + 7) / 8;
        }
        // Upper bound of the forward orientation's score: the alignment along diagonal 0 (one mismatch per differing
        // column) closed by one gap over the length difference (1 + length under the orientation penalties).  Below the
        // reverse orientation's lower bound the forward one wins whatever the two scores are exactly -- the reverse
        // complement is chosen only at a strictly lower score -- and neither aligner runs; ori_fwd then holds the bound
        // (it only orders the queue).  Substitution-only families (C2, C4) decide almost every pair here.
        bool decided = false;
        if (lb_rev > 0) {
            const int n = min(plen, tlen), nw = (n + 15) >> 4;
            const LP Pf = (LP)(lds_seq + 1);
            int ham = 0;
            for (int i = lane; i < nw; i += 64) {
                uint32_t x = Pf[i] ^ T[i];
                x = (x | (x >> 1)) & 0x55555555u;
                const int rem = n - (i << 4);
                if (rem < 16) x &= (1u << (rem << 1)) - 1u;
                ham += __popc(x);
            }
            for (int o = 32; o > 0; o >>= 1) ham += __shfl_xor(ham, o, 64);
            const int ub = ham + (plen != tlen ? 1 + abs(plen - tlen) : 0);
            if (ub < lb_rev) { fwd = ub; decided = true; }
        }
#endif
        bool rev_on = lb_rev <= 0, catching = false;
        int s0_main = 0, s0_catch = 0;
#if SR_SYMBITS == 2
        if (!decided)
#endif
        for (;;) {
            if (!catching && !rev_on && s0_main + OB - 1 >= lb_rev) {
                if (s0_main > 0) { catching = true; s0_catch = 0; } else rev_on = true;
            }
            const int s0 = catching ? s0_catch : s0_main;
            const int jobmask = catching ? 2 : (rev_on ? 3 : 1);
            const int par = (s0 / OB) & 1;
            // rows of the previous block (NULL rows before the first one) and of this one
            GP<OT> pM1 = s0 ? ring + (size_t)((1 - par) * 4 + 0) * w : nul, pM2 = s0 ? ring + (size_t)((1 - par) * 4 + 1) * w : nul;
            GP<OT> pI = s0 ? ring + (size_t)((1 - par) * 4 + 2) * w : nul, pD = s0 ? ring + (size_t)((1 - par) * 4 + 3) * w : nul;
            GP<OT> oM1 = ring + (size_t)(par * 4 + 0) * w, oM2 = ring + (size_t)(par * 4 + 1) * w;
            GP<OT> oI = ring + (size_t)(par * 4 + 2) * w, oD = ring + (size_t)(par * 4 + 3) * w;
            const int Rl = reach(pen, s0 + OB - 1, SR_C_M);
            const int wlo = max(-plen - 1, -Rl - OB - 2), whi = min(tlen + 1, Rl + OB + 2);
            const int glo = (wlo + shift) >> 2, ghi = (whi + shift) >> 2;
            const int nt = (ghi - glo + 60) / 60;
            int hitl[2] = {OB, OB};                              // first level of the block at which the aligner reached the end
#pragma unroll
            for (int job = 0; job < 2; job++) {
                if (!((jobmask >> job) & 1)) continue;
                const LP P = (LP)(lds_seq + (job ? a.max_words : 0) + 1);
                const int base = job * width;
                for (int ti = 0; ti < nt; ti++) {
                    const int g = glo + ti * 60 + lane - 2;
                    const bool owned = (lane >= 2) && (lane < 62) && (g <= ghi);
                    const int k0 = (g << 2) - shift;
                    const unsigned idx0 = (unsigned)(base + (g << 2));
                    const V4<OT> v1 = ld4<OT>(pM1, idx0), v2 = ld4<OT>(pM2, idx0), vi = ld4<OT>(pI, idx0), vd = ld4<OT>(pD, idx0);
                    int m1[4], m2[4], i1[4], d1[4];
                    unsigned lim[4];
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        m1[qq] = (int)v1[qq]; m2[qq] = (int)v2[qq]; i1[qq] = (int)vi[qq]; d1[qq] = (int)vd[qq];
                        lim[qq] = (unsigned)max(min(tlen, plen + k0 + qq), -1);
                    }
                    // LDS-wide symbol / bit index of P[-k0] and T[0]
                    const int cp0 = -k0 + (int)(((uint32_t)(uintptr_t)P >> 2) << SR_WIN_LOG), ct0 = (int)(((uint32_t)(uintptr_t)T >> 2) << SR_WIN_LOG);
                    const int cpb = cp0 << SR_SYM_LOG, ctb = ct0 << SR_SYM_LOG;
                    bool hit_now = false;
                    int hit_at = OB;
#pragma unroll
                    for (int j = 0; j < OB; j++) {
                        const int s = s0 + j;
                        const int R = reach(pen, s, SR_C_M);
                        const int klo = max(-plen, -R), khi = min(tlen, R);
                        // sources: the k-1 / k+1 cells of max(M[s-2], I[s-1]) and max(M[s-2], D[s-1])
                        int ti_[4], td_[4];
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) { ti_[qq] = max(m2[qq], i1[qq]); td_[qq] = max(m2[qq], d1[qq]); }
                        const int tiL = o_lane_left(ti_[3]), tdR = o_lane_right(td_[0]);
                        int mv[4], iv[4], dv[4];
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) {
                            const int k = k0 + qq;
                            const bool inr = (k >= klo) && (k <= khi);
                            int in_ = bnd(((qq == 0) ? tiL : ti_[qq == 0 ? 0 : qq - 1]) + 1, lim[qq]);
                            int dn_ = bnd((qq == 3) ? tdR : td_[qq == 3 ? 3 : qq + 1], lim[qq]);
                            int m = bnd(m1[qq] + 1, lim[qq]);
                            m = max(m, max(in_, dn_));
                            if (!inr) { m = NULLV; in_ = NULLV; dn_ = NULLV; }
                            if (s == 0) { m = (inr && k == 0) ? 0 : NULLV; in_ = NULLV; dn_ = NULLV; }
                            mv[qq] = m; iv[qq] = in_; dv[qq] = dn_;
                        }
                        // extension of every cell of the wave (halo lanes feed owned cells of later levels).  The eight
                        // window reads are in flight together, each addressed by its LDS-wide bit index (the read address
                        // >> 3 and v_alignbit's shift at once); a NULL cell runs through the same code (its reads land
                        // anywhere in or outside the LDS allocation and return what is there or 0, its value stays negative
                        // and is reset by bnd() at the next level), only the "longer than a window" flag looks at validity.
                        unsigned long long pend[4];
                        {
                            uint32_t pl[4], ph[4], tl[4], th[4];
                            int bp[4], bt[4];
#pragma unroll
                            for (int qq = 0; qq < 4; qq++) {
                                bp[qq] = (mv[qq] << SR_SYM_LOG) + cpb - (qq << SR_SYM_LOG); bt[qq] = (mv[qq] << SR_SYM_LOG) + ctb;
                                win_words_bit(bp[qq], pl[qq], ph[qq]); win_words_bit(bt[qq], tl[qq], th[qq]);
                            }
                            asm volatile("; 8 windows in flight" : "+v"(pl[0]), "+v"(ph[0]), "+v"(pl[1]), "+v"(ph[1]), "+v"(pl[2]), "+v"(ph[2]), "+v"(pl[3]), "+v"(ph[3]),
                                                                    "+v"(tl[0]), "+v"(th[0]), "+v"(tl[1]), "+v"(th[1]), "+v"(tl[2]), "+v"(th[2]), "+v"(tl[3]), "+v"(th[3]));
#pragma unroll
                            for (int qq = 0; qq < 4; qq++) {
                                const int nn = (int)lim[qq] - mv[qq];
                                const uint32_t xw = __builtin_amdgcn_alignbit(ph[qq], pl[qq], (uint32_t)bp[qq]) ^
                                                    __builtin_amdgcn_alignbit(th[qq], tl[qq], (uint32_t)bt[qq]);
                                const int ext_ = (int)min3u(ffs_sym(xw), (unsigned)SR_WIN, (unsigned)nn);
                                mv[qq] += ext_;
                                pend[qq] = __builtin_amdgcn_ballot_w64(ext_ == SR_WIN) & __builtin_amdgcn_ballot_w64(mv[qq] >= 0);
                            }
                        }
                        while ((pend[0] | pend[1] | pend[2] | pend[3]) != 0ull) {
#pragma unroll
                            for (int qq = 0; qq < 4; qq++) {
                                if (pend[qq] == 0ull) continue;
                                const int nn = lanes_or_zero(pend[qq], (int)lim[qq] - mv[qq]);      // 0 for the lanes that are done
                                const uint32_t xw = win_sym(mv[qq] + cp0 - qq) ^ win_sym(mv[qq] + ct0);
                                mv[qq] += (int)min3u(ffs_sym(xw), (unsigned)SR_WIN, (unsigned)nn);
                                pend[qq] = __builtin_amdgcn_ballot_w64(xw == 0u && nn > SR_WIN);
                            }
                        }
                        if (!hit_now) {
#pragma unroll
                            for (int qq = 0; qq < 4; qq++)
                                hit_now |= owned && (k0 + qq) == kend && (k0 + qq) >= klo && (k0 + qq) <= khi && mv[qq] >= tlen;
                            if (hit_now) hit_at = j;
                        }
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) { m2[qq] = m1[qq]; m1[qq] = mv[qq]; i1[qq] = iv[qq]; d1[qq] = dv[qq]; }
                    }
                    if (owned) {
                        V4<OT> a1, a2, ai, ad;
#pragma unroll
                        for (int qq = 0; qq < 4; qq++) { a1[qq] = (OT)m1[qq]; a2[qq] = (OT)m2[qq]; ai[qq] = (OT)i1[qq]; ad[qq] = (OT)d1[qq]; }
                        st4<OT>(oM1, idx0, a1); st4<OT>(oM2, idx0, a2); st4<OT>(oI, idx0, ai); st4<OT>(oD, idx0, ad);
                    }
                    // (min over the lanes: exactly one lane owns the end diagonal)
                    const unsigned long long hm = __ballot(hit_now);
                    if (hm) hitl[job] = min(hitl[job], __shfl(hit_at, __ffsll((long long)hm) - 1, 64));
                }
            }
            __syncthreads();                                    // one wave: workgroup-scope fence, the block's rows are visible
            // per level the reference checks the forward aligner first
            const int done = min(hitl[0], hitl[1]);
            const int lv = min(OB - 1, done);
            const unsigned long long nal = (unsigned long long)__popc((unsigned)jobmask);       // aligners this block computed
            for (int j = 0; j <= lv; j++) {                      // level statistics as the level-by-level kernel counts them
                const int R = reach(pen, s0 + j, SR_C_M);
                cells += nal * (unsigned long long)(min(tlen, R) - max(-plen, -R) + 1);
                steps += nal;
            }
            if (catching) {
                // (the forward aligner had not finished before s0_main: a reverse aligner that reaches the end here wins)
                if (hitl[1] < OB) { rev = s0 + hitl[1]; is_rev = 1; break; }
                s0_catch += OB;
                if (s0_catch >= s0_main) { catching = false; rev_on = true; }
                continue;
            }
            if (done < OB) {
                if (hitl[0] <= hitl[1]) fwd = s0 + hitl[0];
                else { rev = s0 + hitl[1]; is_rev = 1; }
                break;
            }
            if (s0 + OB - 1 > smax) { err |= SR_DEV_ERR_SCORE_BOUND; break; }
            s0_main += OB;
        }
        if (lane == 0) {
            a.is_reverse[pair] = is_rev ? 1 : 0;
            a.ori_fwd[pair] = fwd; a.ori_rev[pair] = rev;
        }
    }
    if (lane == 0) {
        if (cells) { atomicAdd(&a.counters[0], cells); atomicAdd(&a.counters[6], cells); }
        if (steps) atomicAdd(&a.counters[1], steps);
        if (err) atomicOr(a.error_flag, err);
    }
}
#undef OB

}  // namespace
using namespace SR_NS;
#if SR_SYMBITS == 2
// One workgroup per sequence: the set of 8-mers (16 bits of the packed forward copy) that occur in it, as 2^16 bits.
// sr_orient_blk_kernel counts how many 8-mer positions of the reverse-complemented query hit the target's set: by the
// q-gram lemma an alignment with d edits leaves at least n - 7 - 8 d of them intact, so the reverse orientation's
// score (>= its edit distance) is at least (positions - hits) / 8.
__global__ void __launch_bounds__(256) sr_kmer_bits_kernel(SrAlignArgs a, uint32_t *kbits) {
    const uint32_t sq = blockIdx.x;
    uint32_t *bm = kbits + (size_t)sq * 2048;
    for (int i = threadIdx.x; i < 2048; i += 256) bm[i] = 0u;
    __syncthreads();
    const int len = (int)a.seqlen[sq];
    const uint32_t *w = a.seqwords + a.word_off_fwd[sq];
    for (int i = threadIdx.x; i + 8 <= len; i += 256) {
        const int wi = i >> 4, sh = (i & 15) << 1;
        const uint64_t v = ((uint64_t)w[wi + 1] << 32) | (uint64_t)w[wi];
        const uint32_t code = (uint32_t)(v >> sh) & 0xffffu;
        atomicOr(&bm[code >> 5], 1u << (code & 31u));
    }
}
extern "C" int srk_kmer_bits(const SrAlignArgs *a, uint32_t nseq, uint32_t *kbits, void *stream) {
    if (nseq == 0) return 0;
    hipLaunchKernelGGL(sr_kmer_bits_kernel, dim3(nseq), dim3(256), 0, (hipStream_t)stream, *a, kbits);
    return (int)hipGetLastError();
}
#endif
extern "C" int SRK_NAME(srk_orient)(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const bool blocked = !a->ori.two && a->ori.x == 1 && a->ori.o1 == 1 && a->ori.e1 == 1 && !getenv("SR_ORIENT_LEVELS");
    if (blocked) {
        if (lds_bytes > 32 * 1024) {
            hipError_t e = off16 ? hipFuncSetAttribute((const void *)sr_orient_blk_kernel<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
                                 : hipFuncSetAttribute((const void *)sr_orient_blk_kernel<int32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return (int)e;
        }
        if (off16) hipLaunchKernelGGL((sr_orient_blk_kernel<int16_t>), dim3(nwg), dim3(64), lds_bytes, st, *a);
        else hipLaunchKernelGGL((sr_orient_blk_kernel<int32_t>), dim3(nwg), dim3(64), lds_bytes, st, *a);
        return (int)hipGetLastError();
    }
    if (off16) {
        if (lds_bytes > 32 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)sr_orient_kernel<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((sr_orient_kernel<int16_t>), dim3(nwg), dim3(64), lds_bytes, st, *a);
    } else {
        if (lds_bytes > 32 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)sr_orient_kernel<int32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL((sr_orient_kernel<int32_t>), dim3(nwg), dim3(64), lds_bytes, st, *a);
    }
    return (int)hipGetLastError();
}
