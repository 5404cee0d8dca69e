/* ORACLE (test infrastructure) -- gap-affine / gap-affine-2p wavefront
 * alignment, end-to-end, exact (no heuristic), restating WFA2-lib as driven by
 * the reference through lib_wfa2 (src/wfa.rs:41-75:
 * AffineWavefronts::with_penalties_affine2p_and_memory_mode(.., Ultralow),
 * AlignmentScope::Alignment, AlignmentSpan::End2End, HeuristicStrategy::None;
 * .align/.score/.cigar usage src/seqrush_bidirected.rs:218-236).
 *
 * WFA2-lib / lib_wfa2 @819b82cd are NOT in the reference tree, so
 * this file restates the published algorithm:
 *   - wavefront recurrences, diagonal k = h - v, offset = h (text position),
 *     pattern = query (v), text = target (h);
 *   - backtrace: among predecessor candidates take the maximum offset, ties
 *     resolved by the piggy-backed type tag, priority high->low:
 *     mismatch > D2-ext > D2-open > D1-ext > D1-open > I2-ext > I2-open >
 *     I1-ext > I1-open   (SURVEY Appendix A);
 *   - MemoryMode::Ultralow = biWFA: forward and reverse score-only aligners
 *     advance alternately, overlap detection per component, recursion on the
 *     breakpoint with begin/end component constraints, fall back to plain WFA
 *     with backtrace when the remaining score <= 250 or both lengths <= 100.
 * Known answers that pin it: tests/test_wfa2_cigar_debug.rs:4-30
 * (ATCGATCG vs ATCGATCGATCG -> MMMMMMMMIIII), tests/test_cigar_validity.rs.
 * Co-optimal tie-breaking and biWFA breakpoint choice vs the real WFA2-lib:
 * PARITY UNPINNED.  The HIP kernels implement exactly the rules written here.
 *
 * Cell semantics used here and on the GPU: every stored offset is either
 * WF_NULL or an in-bounds furthest-reaching point (0 <= h <= tlen,
 * 0 <= v <= plen); out-of-bounds candidates are nulled for all components.
 */
#include "sr_oracle.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define WF_NULL (-(1 << 28))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))

/* backtrace type tags (priority = numeric value) */
enum { BT_I1_OPEN = 1, BT_I1_EXT = 2, BT_I2_OPEN = 3, BT_I2_EXT = 4,
       BT_D1_OPEN = 5, BT_D1_EXT = 6, BT_D2_OPEN = 7, BT_D2_EXT = 8,
       BT_MISMS = 9 };

typedef struct {
    const uint8_t *p, *t;
    int plen, tlen;
    int rev;
} view_t;

static inline int base_eq(const view_t *w, int v, int h) {
    if (w->rev) return w->p[w->plen - 1 - v] == w->t[w->tlen - 1 - h];
    return w->p[v] == w->t[h];
}

typedef struct {
    int score;      /* score held by this slot, -1 = none */
    int lo, hi;     /* lo > hi : empty level */
    int off0;       /* index = k + off0 */
    int wlo, whi;   /* modular rows: diagonals written by the level the slot held last */
    int ilo, ihi;   /* modular rows: diagonals initialised so far; inside, everything outside [wlo, whi] is WF_NULL */
    int32_t *c[5];
} level_t;

typedef struct {
    view_t w;
    int x, o1, e1, o2, e2, two;
    int scope;
    int modular;
    level_t *lv;
    int nlv;        /* modular: scope ; full: allocated count */
    int cap, shift; /* modular storage */
    int32_t *nullrow; /* modular: a row of WF_NULL standing in for absent source levels */
    int null_ilo, null_ihi;
    uint64_t cells;
} wfa_t;

static _Thread_local uint64_t g_last_cells;
uint64_t sro_wfa_last_cells(void) { return g_last_cells; }

/* Per-thread bump arena for the wavefront rows: one aligner object per thread like the reference's rayon workers
 * (no malloc per level; VERDICT r1 "per-thread arenas").  Chunks are kept for the thread's lifetime; a computation
 * takes a mark on entry and releases to it on exit (the biWFA recursion frees a level's aligners before it
 * descends, so marks nest). */
typedef struct arena_chunk { struct arena_chunk *next; size_t cap, top; } arena_chunk;
typedef struct { arena_chunk *c; size_t top; } arena_mark;
static _Thread_local arena_chunk *g_arena_head, *g_arena_cur;
static void *arena_alloc(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    for (;;) {
        if (g_arena_cur && g_arena_cur->top + bytes <= g_arena_cur->cap) {
            void *p = (char *)(g_arena_cur + 1) + g_arena_cur->top;
            g_arena_cur->top += bytes;
            return p;
        }
        if (g_arena_cur && g_arena_cur->next) { g_arena_cur = g_arena_cur->next; g_arena_cur->top = 0; continue; }
        size_t cap = (size_t)8 << 20;
        if (cap < bytes) cap = bytes;
        arena_chunk *n = (arena_chunk *)malloc(sizeof(arena_chunk) + 64 + cap);
        n->next = NULL; n->cap = cap; n->top = 0;
        if (g_arena_cur) g_arena_cur->next = n; else g_arena_head = n;
        g_arena_cur = n;
    }
}
static arena_mark arena_get_mark(void) {
    if (!g_arena_cur) (void)arena_alloc(64);
    arena_mark m = { g_arena_cur, g_arena_cur->top };
    return m;
}
static void arena_release(arena_mark m) { g_arena_cur = m.c; g_arena_cur->top = m.top; }

static int pen_scope(const sro_penalties *pen) {
    int s = IMAX(pen->mismatch, pen->gap_open1 + pen->gap_ext1);
    if (pen->gap_open2 >= 0) s = IMAX(s, pen->gap_open2 + pen->gap_ext2);
    return s + 1;
}

static void wfa_init(wfa_t *a, const view_t *w, const sro_penalties *pen,
                     int modular) {
    memset(a, 0, sizeof(*a));
    a->w = *w;
    a->x = pen->mismatch; a->o1 = pen->gap_open1; a->e1 = pen->gap_ext1;
    a->two = pen->gap_open2 >= 0;
    a->o2 = a->two ? pen->gap_open2 : 0;
    a->e2 = a->two ? pen->gap_ext2 : 0;
    a->scope = pen_scope(pen);
    a->modular = modular;
    if (modular) {
        a->nlv = a->scope;
        a->cap = w->plen + w->tlen + 3;
        a->shift = w->plen + 1;
        a->lv = (level_t *)arena_alloc((size_t)a->nlv * sizeof(level_t));
        memset(a->lv, 0, (size_t)a->nlv * sizeof(level_t));
        for (int i = 0; i < a->nlv; i++) {
            a->lv[i].score = -1;
            a->lv[i].lo = 1; a->lv[i].hi = 0;
            a->lv[i].off0 = a->shift;
            a->lv[i].wlo = 1; a->lv[i].whi = 0; a->lv[i].ilo = 1; a->lv[i].ihi = 0;
            for (int j = 0; j < 5; j++)
                a->lv[i].c[j] = (int32_t *)arena_alloc(sizeof(int32_t) * (size_t)a->cap);
        }
        /* (rows are initialised to WF_NULL window by window as the wavefront widens: row_cover) */
        a->nullrow = (int32_t *)arena_alloc(sizeof(int32_t) * (size_t)a->cap);
        a->null_ilo = 1; a->null_ihi = 0;
    } else {
        a->nlv = 0;
        a->lv = NULL;
    }
}

static void wfa_free(wfa_t *a) {       /* rows live in the thread's arena: the caller releases its mark */
    a->lv = NULL;
}

/* slot that will hold score s (allocating in full mode) */
static level_t *wfa_slot(wfa_t *a, int s) {
    if (a->modular) return &a->lv[s % a->scope];
    if (s >= a->nlv) {
        int n = a->nlv ? a->nlv : 64;
        while (n <= s) n *= 2;
        level_t *nl = (level_t *)arena_alloc(sizeof(level_t) * (size_t)n);
        if (a->nlv) memcpy(nl, a->lv, sizeof(level_t) * (size_t)a->nlv);
        a->lv = nl;
        for (int i = a->nlv; i < n; i++) {
            memset(&a->lv[i], 0, sizeof(level_t));
            a->lv[i].score = -1; a->lv[i].lo = 1; a->lv[i].hi = 0;
        }
        a->nlv = n;
    }
    return &a->lv[s];
}

/* level holding score s, or NULL when absent/empty */
static const level_t *wfa_level(const wfa_t *a, int s) {
    if (s < 0) return NULL;
    const level_t *L;
    if (a->modular) L = &a->lv[s % a->scope];
    else { if (s >= a->nlv) return NULL; L = &a->lv[s]; }
    if (L->score != s || L->lo > L->hi) return NULL;
    return L;
}

static inline int32_t lv_get(const level_t *L, int comp, int k) {
    if (!L || k < L->lo || k > L->hi || !L->c[comp]) return WF_NULL;
    return L->c[comp][k + L->off0];
}

static void lv_prepare(wfa_t *a, level_t *L, int s, int lo, int hi) {
    L->score = s; L->lo = lo; L->hi = hi;
    if (!a->modular) {
        L->off0 = -lo;
        size_t n = (size_t)(hi >= lo ? hi - lo + 1 : 1);
        for (int j = 0; j < 5; j++) L->c[j] = (int32_t *)arena_alloc(sizeof(int32_t) * n);
    }
}

static inline int32_t bnd(int32_t c, uint32_t lim) {
    return ((uint32_t)c > lim) ? WF_NULL : c;
}

/* match extension, 8 bytes per compare like WFA2-lib's wavefront_extend (64-bit blocks + count-trailing-zeros) */
static inline int32_t wf_extend(const view_t *w, int k, int32_t off) {
    int v = off - k, h = off;
    const int plen = w->plen, tlen = w->tlen;
    if (!w->rev) {
        const uint8_t *p = w->p + v, *t = w->t + h;
        int n = plen - v < tlen - h ? plen - v : tlen - h, i = 0;
        while (i + 8 <= n) {
            uint64_t a, b;
            memcpy(&a, p + i, 8); memcpy(&b, t + i, 8);
            const uint64_t x = a ^ b;
            if (x) return h + i + (__builtin_ctzll(x) >> 3);
            i += 8;
        }
        while (i < n && p[i] == t[i]) i++;
        return h + i;
    } else {
        const uint8_t *p = w->p + (plen - 1 - v), *t = w->t + (tlen - 1 - h);     /* walk downwards */
        int n = plen - v < tlen - h ? plen - v : tlen - h, i = 0;
        while (i + 8 <= n) {
            uint64_t a, b;
            memcpy(&a, p - i - 7, 8); memcpy(&b, t - i - 7, 8);
            const uint64_t x = a ^ b;
            if (x) return h + i + (__builtin_clzll(x) >> 3);
            i += 8;
        }
        while (i < n && *(p - i) == *(t - i)) i++;
        return h + i;
    }
}

/* level 0: begin component at (k=0, offset 0) */
static void wfa_level0(wfa_t *a, int begin) {
    level_t *L = wfa_slot(a, 0);
    lv_prepare(a, L, 0, 0, 0);
    if (a->modular) { L->wlo = 0; L->whi = 0; L->ilo = 0; L->ihi = 0; }
    for (int j = 0; j < 5; j++) L->c[j][0 + L->off0] = WF_NULL;
    int32_t off = 0;
    if (begin == SRO_M) off = wf_extend(&a->w, 0, 0);
    L->c[begin][0 + L->off0] = off;
    a->cells += 1;
}

/* make the rows of a modular slot (L != NULL) or the NULL row readable over diagonals [lo, hi]: what has never been
 * initialised becomes WF_NULL (windows only grow, a few cells per level) */
static void row_cover(wfa_t *a, level_t *L, int lo, int hi) {
    const int sh = a->shift;
    int *ilo = L ? &L->ilo : &a->null_ilo, *ihi = L ? &L->ihi : &a->null_ihi;
    const int nrows = L ? 5 : 1;
    if (*ilo > *ihi) {
        for (int j = 0; j < nrows; j++) { int32_t *r = (L ? L->c[j] : a->nullrow) + sh; for (int k = lo; k <= hi; k++) r[k] = WF_NULL; }
        *ilo = lo; *ihi = hi;
        return;
    }
    for (int j = 0; j < nrows; j++) {
        int32_t *r = (L ? L->c[j] : a->nullrow) + sh;
        for (int k = lo; k < *ilo; k++) r[k] = WF_NULL;
        for (int k = *ihi + 1; k <= hi; k++) r[k] = WF_NULL;
    }
    if (lo < *ilo) *ilo = lo;
    if (hi > *ihi) *ihi = hi;
}

/* recurrences of one level over [lo, hi]: distinct rows in, distinct rows out (restrict parameters: what lets the
 * compiler vectorise without alias checks) */
static void __attribute__((noinline)) step_kernel(int lo, int hi, int plen, int tlen, int two,
        const int32_t *restrict mx, const int32_t *restrict mo1, const int32_t *restrict si1, const int32_t *restrict sd1,
        const int32_t *restrict mo2, const int32_t *restrict si2, const int32_t *restrict sd2,
        int32_t *restrict oM, int32_t *restrict oI1, int32_t *restrict oI2, int32_t *restrict oD1, int32_t *restrict oD2) {
    const int32_t nul2 = two ? INT32_MAX : -1;               /* one-piece: every I2 / D2 cell fails "<= limit" */
    for (int k = lo; k <= hi; k++) {
        const int32_t limv = IMIN(tlen, plen + k);           /* >= 0 on [-plen, tlen] */
        const int32_t lim2 = IMIN(limv, nul2);
        int32_t i1 = IMAX(mo1[k - 1], si1[k - 1]) + 1;
        int32_t d1 = IMAX(mo1[k + 1], sd1[k + 1]);
        int32_t i2 = IMAX(mo2[k - 1], si2[k - 1]) + 1;
        int32_t d2 = IMAX(mo2[k + 1], sd2[k + 1]);
        int32_t m = mx[k] + 1;
        i1 = ((uint32_t)i1 > (uint32_t)limv) ? WF_NULL : i1;
        d1 = ((uint32_t)d1 > (uint32_t)limv) ? WF_NULL : d1;
        i2 = (lim2 < 0 || (uint32_t)i2 > (uint32_t)lim2) ? WF_NULL : i2;
        d2 = (lim2 < 0 || (uint32_t)d2 > (uint32_t)lim2) ? WF_NULL : d2;
        m = ((uint32_t)m > (uint32_t)limv) ? WF_NULL : m;
        m = IMAX(m, IMAX(IMAX(i1, i2), IMAX(d1, d2)));
        oM[k] = m; oI1[k] = i1; oI2[k] = i2; oD1[k] = d1; oD2[k] = d2;
    }
}

/* The same step for the modular (score-only, biWFA) aligners, laid out the way WFA2-lib's own compute kernels are:
 * every row of a slot spans all diagonals and holds WF_NULL outside the level it stores, absent source levels read a
 * NULL row, so the recurrences of a level are one branch-free loop over [lo, hi] the compiler vectorises; the match
 * extension follows as a second loop.  Cell for cell the values of wfa_step (tests/test_oracle_golden.py compares the
 * two on random inputs through SRO_ORACLE_SCALAR_STEP=1). */
static int g_scalar_step = -1;
static void wfa_step_scalar(wfa_t *a, int s);
static void wfa_step_modular(wfa_t *a, int s) {
    const level_t *Lx = wfa_level(a, s - a->x);
    const level_t *Lo1 = wfa_level(a, s - a->o1 - a->e1);
    const level_t *Le1 = wfa_level(a, s - a->e1);
    const level_t *Lo2 = a->two ? wfa_level(a, s - a->o2 - a->e2) : NULL;
    const level_t *Le2 = a->two ? wfa_level(a, s - a->e2) : NULL;
    int lo = INT_MAX, hi = INT_MIN;
    if (Lx) { lo = IMIN(lo, Lx->lo); hi = IMAX(hi, Lx->hi); }
    if (Lo1) { lo = IMIN(lo, Lo1->lo - 1); hi = IMAX(hi, Lo1->hi + 1); }
    if (Le1) { lo = IMIN(lo, Le1->lo - 1); hi = IMAX(hi, Le1->hi + 1); }
    if (Lo2) { lo = IMIN(lo, Lo2->lo - 1); hi = IMAX(hi, Lo2->hi + 1); }
    if (Le2) { lo = IMIN(lo, Le2->lo - 1); hi = IMAX(hi, Le2->hi + 1); }
    level_t *L = wfa_slot(a, s);
    const int plen = a->w.plen, tlen = a->w.tlen;
    if (lo <= hi) { lo = IMAX(lo, -plen); hi = IMIN(hi, tlen); }
    /* every row the loop reads or writes is initialised over [lo - 1, hi + 1] */
    if (lo <= hi) {
        const int clo = lo - 1, chi = hi + 1;
        row_cover(a, (level_t *)Lx, clo, chi); row_cover(a, (level_t *)Lo1, clo, chi); row_cover(a, (level_t *)Le1, clo, chi);
        if (Lo2) row_cover(a, (level_t *)Lo2, clo, chi);
        if (Le2) row_cover(a, (level_t *)Le2, clo, chi);
        row_cover(a, NULL, clo, chi);
        row_cover(a, L, clo, chi);
    }
    /* the slot's old level (s - scope) goes: its cells outside the new range become WF_NULL again */
    const int sh = a->shift;
    {
        const int nlo = lo <= hi ? lo : INT_MAX, nhi = lo <= hi ? hi : INT_MIN;
        for (int k = L->wlo; k <= L->whi; k++)
            if (k < nlo || k > nhi) for (int j = 0; j < 5; j++) L->c[j][k + sh] = WF_NULL;
    }
    if (lo > hi) { L->score = s; L->lo = 1; L->hi = 0; L->wlo = 1; L->whi = 0; return; }
    L->score = s; L->wlo = lo; L->whi = hi;
    const int32_t *restrict mx = (Lx ? Lx->c[SRO_M] : a->nullrow) + sh;
    const int32_t *restrict mo1 = (Lo1 ? Lo1->c[SRO_M] : a->nullrow) + sh;
    const int32_t *restrict si1 = (Le1 ? Le1->c[SRO_I1] : a->nullrow) + sh;
    const int32_t *restrict sd1 = (Le1 ? Le1->c[SRO_D1] : a->nullrow) + sh;
    const int32_t *restrict mo2 = (Lo2 ? Lo2->c[SRO_M] : a->nullrow) + sh;
    const int32_t *restrict si2 = (Le2 ? Le2->c[SRO_I2] : a->nullrow) + sh;
    const int32_t *restrict sd2 = (Le2 ? Le2->c[SRO_D2] : a->nullrow) + sh;
    int32_t *restrict oM = L->c[SRO_M] + sh, *restrict oI1 = L->c[SRO_I1] + sh, *restrict oI2 = L->c[SRO_I2] + sh;
    int32_t *restrict oD1 = L->c[SRO_D1] + sh, *restrict oD2 = L->c[SRO_D2] + sh;
    step_kernel(lo, hi, plen, tlen, a->two, mx, mo1, si1, sd1, mo2, si2, sd2, oM, oI1, oI2, oD1, oD2);
    int any_lo = INT_MAX, any_hi = INT_MIN;
    for (int k = lo; k <= hi; k++) {
        const int32_t m = oM[k];
        if (m < 0) continue;
        oM[k] = wf_extend(&a->w, k, m);
        if (any_lo == INT_MAX) any_lo = k;
        any_hi = k;
    }
    a->cells += (uint64_t)(hi - lo + 1);
    if (any_lo > any_hi) { L->lo = 1; L->hi = 0; }
    else { L->lo = any_lo; L->hi = any_hi; }
}

/* one score step: compute + bound + extend (WFA2 wavefront_compute_affine2p
 * + wavefront_extend_end2end) */
static void wfa_step(wfa_t *a, int s) {
    if (a->modular) {
        if (g_scalar_step < 0) { const char *e = getenv("SRO_ORACLE_SCALAR_STEP"); g_scalar_step = (e && atoi(e)) ? 1 : 0; }
        if (!g_scalar_step) { wfa_step_modular(a, s); return; }
    }
    wfa_step_scalar(a, s);
}
static void wfa_step_scalar(wfa_t *a, int s) {
    const level_t *Lx = wfa_level(a, s - a->x);
    const level_t *Lo1 = wfa_level(a, s - a->o1 - a->e1);
    const level_t *Le1 = wfa_level(a, s - a->e1);
    const level_t *Lo2 = a->two ? wfa_level(a, s - a->o2 - a->e2) : NULL;
    const level_t *Le2 = a->two ? wfa_level(a, s - a->e2) : NULL;
    int lo = INT_MAX, hi = INT_MIN;
    if (Lx) { lo = IMIN(lo, Lx->lo); hi = IMAX(hi, Lx->hi); }
    if (Lo1) { lo = IMIN(lo, Lo1->lo - 1); hi = IMAX(hi, Lo1->hi + 1); }
    if (Le1) { lo = IMIN(lo, Le1->lo - 1); hi = IMAX(hi, Le1->hi + 1); }
    if (Lo2) { lo = IMIN(lo, Lo2->lo - 1); hi = IMAX(hi, Lo2->hi + 1); }
    if (Le2) { lo = IMIN(lo, Le2->lo - 1); hi = IMAX(hi, Le2->hi + 1); }
    level_t *L = wfa_slot(a, s);
    if (lo > hi) { L->score = s; L->lo = 1; L->hi = 0; return; }
    const int plen = a->w.plen, tlen = a->w.tlen;
    lo = IMAX(lo, -plen); hi = IMIN(hi, tlen);
    if (lo > hi) { L->score = s; L->lo = 1; L->hi = 0; return; }
    /* in modular mode the output slot may alias a source level only when
     * that source is older than scope, which never happens */
    lv_prepare(a, L, s, lo, hi);
    int any_lo = INT_MAX, any_hi = INT_MIN;
    for (int k = lo; k <= hi; k++) {
        const uint32_t lim = (uint32_t)IMIN(tlen, plen + k);
        /* every candidate is bounded on its own, so M is null iff all
         * components are null on this diagonal */
        int32_t i1 = bnd(IMAX(lv_get(Lo1, SRO_M, k - 1), lv_get(Le1, SRO_I1, k - 1)) + 1, lim);
        int32_t d1 = bnd(IMAX(lv_get(Lo1, SRO_M, k + 1), lv_get(Le1, SRO_D1, k + 1)), lim);
        int32_t i2 = WF_NULL, d2 = WF_NULL;
        if (a->two) {
            i2 = bnd(IMAX(lv_get(Lo2, SRO_M, k - 1), lv_get(Le2, SRO_I2, k - 1)) + 1, lim);
            d2 = bnd(IMAX(lv_get(Lo2, SRO_M, k + 1), lv_get(Le2, SRO_D2, k + 1)), lim);
        }
        int32_t m = bnd(lv_get(Lx, SRO_M, k) + 1, lim);
        m = IMAX(m, IMAX(IMAX(i1, i2), IMAX(d1, d2)));
        if (m < 0) m = WF_NULL;
        if (m >= 0) m = wf_extend(&a->w, k, m);
        const int idx = k + L->off0;
        L->c[SRO_M][idx] = m; L->c[SRO_I1][idx] = i1; L->c[SRO_I2][idx] = i2;
        L->c[SRO_D1][idx] = d1; L->c[SRO_D2][idx] = d2;
        if (m >= 0) { any_lo = IMIN(any_lo, k); any_hi = IMAX(any_hi, k); }
    }
    a->cells += (uint64_t)(hi - lo + 1);
    /* M is the max of all components, so a level is empty iff M is null
     * everywhere; trimming null ends changes no value */
    if (any_lo > any_hi) { L->lo = 1; L->hi = 0; }
    else if (a->modular) { L->lo = any_lo; L->hi = any_hi; }
    else {
        /* full mode keeps storage base at the original lo */
        L->off0 = -lo; L->lo = any_lo; L->hi = any_hi;
    }
}

/* ---------------- CIGAR buffer ---------------- */
typedef struct { uint8_t *b; int n, cap; } cig_t;
static void cig_push(cig_t *c, uint8_t op, int count) {
    if (count <= 0) return;
    if (c->n + count > c->cap) {
        int nc = c->cap ? c->cap : 256;
        while (nc < c->n + count) nc *= 2;
        c->b = (uint8_t *)realloc(c->b, (size_t)nc);
        c->cap = nc;
    }
    memset(c->b + c->n, op, (size_t)count);
    c->n += count;
}

/* ---------------- plain WFA with backtrace ---------------- */
static int bt_best(int32_t *best_off, int *best_type, int32_t off, int type) {
    if (off < 0) return 0;
    if (off > *best_off || (off == *best_off && type > *best_type)) {
        *best_off = off; *best_type = type;
    }
    return 1;
}

/* Full-memory WFA on view w between components cb -> ce.  Appends the raw
 * CIGAR (forward order) to out.  Returns score or <0 on failure. */
static int wfa_full(const view_t *w, const sro_penalties *pen, int cb, int ce,
                    cig_t *out, uint64_t *cells) {
    wfa_t a;
    const arena_mark mark = arena_get_mark();
    wfa_init(&a, w, pen, 0);
    const int plen = w->plen, tlen = w->tlen;
    const int k_end = tlen - plen;
    /* hard bound: delete everything + insert everything (+ slack) */
    long smax = (long)a.o1 + (long)a.e1 * plen + (long)a.o1 + (long)a.e1 * tlen
                + 4L * (a.two ? IMAX(a.o1, a.o2) : a.o1) + 16;
    wfa_level0(&a, cb);
    int s = 0, found = 0;
    for (;;) {
        const level_t *L = wfa_level(&a, s);
        if (L && lv_get(L, ce, k_end) >= tlen) { found = 1; break; }
        if (s > smax) break;
        s++;
        wfa_step(&a, s);
    }
    if (!found) { wfa_free(&a); arena_release(mark); return -1; }
    const int score = s;
    /* backtrace (WFA2 wavefront_backtrace_affine semantics) */
    cig_t r = {0, 0, 0};   /* reversed */
    int k = k_end, comp = ce;
    int32_t off = tlen;
    int ok = 1;
    for (;;) {
        if (comp == SRO_M) {
            if (s == 0) {
                /* only reachable when cb == M: initial stroke of matches */
                if (cb != SRO_M || k != 0) { ok = 0; break; }
                cig_push(&r, 'M', off);
                off = 0;
                break;
            }
            int32_t bo = WF_NULL; int bty = 0;
            const level_t *Lx = wfa_level(&a, s - a.x);
            const level_t *Lo1 = wfa_level(&a, s - a.o1 - a.e1);
            const level_t *Le1 = wfa_level(&a, s - a.e1);
            const uint32_t lim = (uint32_t)IMIN(tlen, plen + k);
            bt_best(&bo, &bty, bnd(lv_get(Lx, SRO_M, k) + 1, lim), BT_MISMS);
            bt_best(&bo, &bty, bnd(lv_get(Lo1, SRO_M, k - 1) + 1, lim), BT_I1_OPEN);
            bt_best(&bo, &bty, bnd(lv_get(Le1, SRO_I1, k - 1) + 1, lim), BT_I1_EXT);
            bt_best(&bo, &bty, bnd(lv_get(Lo1, SRO_M, k + 1), lim), BT_D1_OPEN);
            bt_best(&bo, &bty, bnd(lv_get(Le1, SRO_D1, k + 1), lim), BT_D1_EXT);
            if (a.two) {
                const level_t *Lo2 = wfa_level(&a, s - a.o2 - a.e2);
                const level_t *Le2 = wfa_level(&a, s - a.e2);
                bt_best(&bo, &bty, bnd(lv_get(Lo2, SRO_M, k - 1) + 1, lim), BT_I2_OPEN);
                bt_best(&bo, &bty, bnd(lv_get(Le2, SRO_I2, k - 1) + 1, lim), BT_I2_EXT);
                bt_best(&bo, &bty, bnd(lv_get(Lo2, SRO_M, k + 1), lim), BT_D2_OPEN);
                bt_best(&bo, &bty, bnd(lv_get(Le2, SRO_D2, k + 1), lim), BT_D2_EXT);
            }
            if (bty == 0 || bo > off) { ok = 0; break; }
            cig_push(&r, 'M', off - bo);
            off = bo;
            switch (bty) {
            case BT_MISMS: cig_push(&r, 'X', 1); off -= 1; s -= a.x; break;
            case BT_I1_OPEN: cig_push(&r, 'I', 1); off -= 1; k -= 1; s -= a.o1 + a.e1; break;
            case BT_I1_EXT: cig_push(&r, 'I', 1); off -= 1; k -= 1; s -= a.e1; comp = SRO_I1; break;
            case BT_I2_OPEN: cig_push(&r, 'I', 1); off -= 1; k -= 1; s -= a.o2 + a.e2; break;
            case BT_I2_EXT: cig_push(&r, 'I', 1); off -= 1; k -= 1; s -= a.e2; comp = SRO_I2; break;
            case BT_D1_OPEN: cig_push(&r, 'D', 1); k += 1; s -= a.o1 + a.e1; break;
            case BT_D1_EXT: cig_push(&r, 'D', 1); k += 1; s -= a.e1; comp = SRO_D1; break;
            case BT_D2_OPEN: cig_push(&r, 'D', 1); k += 1; s -= a.o2 + a.e2; break;
            case BT_D2_EXT: cig_push(&r, 'D', 1); k += 1; s -= a.e2; comp = SRO_D2; break;
            }
        } else {
            if (s == 0) {
                /* begin component reached at the origin */
                if (comp != cb || k != 0 || off != 0) ok = 0;
                break;
            }
            const int is_ins = (comp == SRO_I1 || comp == SRO_I2);
            const int o = (comp == SRO_I1 || comp == SRO_D1) ? a.o1 : a.o2;
            const int e = (comp == SRO_I1 || comp == SRO_D1) ? a.e1 : a.e2;
            const level_t *Lo = wfa_level(&a, s - o - e);
            const level_t *Le = wfa_level(&a, s - e);
            int32_t c_open, c_ext;
            const uint32_t lim = (uint32_t)IMIN(tlen, plen + k);
            if (is_ins) {
                c_open = bnd(lv_get(Lo, SRO_M, k - 1) + 1, lim);
                c_ext = bnd(lv_get(Le, comp, k - 1) + 1, lim);
            } else {
                c_open = bnd(lv_get(Lo, SRO_M, k + 1), lim);
                c_ext = bnd(lv_get(Le, comp, k + 1), lim);
            }
            /* ext tag outranks open tag */
            int take_ext;
            if (c_ext >= 0 && c_ext >= c_open) take_ext = 1;
            else if (c_open >= 0) take_ext = 0;
            else { ok = 0; break; }
            if ((take_ext ? c_ext : c_open) != off) { ok = 0; break; }
            if (is_ins) { cig_push(&r, 'I', 1); off -= 1; k -= 1; }
            else { cig_push(&r, 'D', 1); k += 1; }
            if (take_ext) s -= e; else { s -= o + e; comp = SRO_M; }
        }
        if (s < 0) { ok = 0; break; }
    }
    if (ok) {
        for (int i = r.n - 1; i >= 0; i--) cig_push(out, r.b[i], 1);
    }
    free(r.b);
    if (cells) *cells += a.cells;
    wfa_free(&a);
    arena_release(mark);
    return ok ? score : -2;
}

/* ---------------- biWFA ---------------- */
typedef struct {
    int score, score_f, score_r, k_f, k_r, comp;
    int32_t off_f, off_r;
} bp_t;

static int level_max_ak(const level_t *L) {
    int mx = 0;
    if (!L) return 0;
    const int32_t *restrict m = L->c[SRO_M] + L->off0;
    for (int k = L->lo; k <= L->hi; k++) {
        const int32_t o = m[k];
        const int ak = o < 0 ? 0 : 2 * o - k;
        mx = ak > mx ? ak : mx;
    }
    return mx;
}

/* breakpoint test on one component pair (m2m / indel2indel) */
static void bp_check(const level_t *L0, const level_t *L1, int comp, int plen,
                     int tlen, int score_0, int score_1, int gap_open,
                     int bp_forward, bp_t *bp) {
    /* L0, L1 non-null levels */
    const int kinv = tlen - plen;
    int lo_0 = L0->lo, hi_0 = L0->hi;
    int lo_1 = kinv - L1->hi, hi_1 = kinv - L1->lo;
    if (hi_1 < lo_0 || hi_0 < lo_1) return;
    int min_hi = IMIN(hi_0, hi_1), max_lo = IMAX(lo_0, lo_1);
    for (int k_0 = max_lo; k_0 <= min_hi; k_0++) {
        const int k_1 = kinv - k_0;
        const int32_t o0 = L0->c[comp][k_0 + L0->off0];
        const int32_t o1 = L1->c[comp][k_1 + L1->off0];
        if (o0 < 0 || o1 < 0) continue;   /* stored values are in-bounds or null */
        if (o0 + o1 >= tlen && score_0 + score_1 - gap_open < bp->score) {
            if (bp_forward) {
                bp->score_f = score_0; bp->score_r = score_1;
                bp->k_f = k_0; bp->k_r = k_1; bp->off_f = o0; bp->off_r = o1;
            } else {
                bp->score_f = score_1; bp->score_r = score_0;
                bp->k_f = k_1; bp->k_r = k_0; bp->off_f = o1; bp->off_r = o0;
            }
            bp->score = score_0 + score_1 - gap_open;
            bp->comp = comp;
            return;
        }
    }
}

static void bialign_overlap(const wfa_t *a0, const wfa_t *a1, int score_0,
                            int score_1, int bp_forward, bp_t *bp) {
    const level_t *L0 = wfa_level(a0, score_0);
    if (!L0) return;
    const int plen = a0->w.plen, tlen = a0->w.tlen;
    for (int i = 0; i < a0->scope; i++) {
        const int score_i = score_1 - i;
        if (score_i < 0) break;
        const level_t *Li = wfa_level(a1, score_i);
        if (a0->two && Li) {
            if (score_0 + score_i - a0->o2 < bp->score) {
                bp_check(L0, Li, SRO_D2, plen, tlen, score_0, score_i, a0->o2, bp_forward, bp);
                bp_check(L0, Li, SRO_I2, plen, tlen, score_0, score_i, a0->o2, bp_forward, bp);
            }
        }
        if (Li && score_0 + score_i - a0->o1 < bp->score) {
            bp_check(L0, Li, SRO_D1, plen, tlen, score_0, score_i, a0->o1, bp_forward, bp);
            bp_check(L0, Li, SRO_I1, plen, tlen, score_0, score_i, a0->o1, bp_forward, bp);
        }
        if (score_0 + score_i >= bp->score) continue;
        if (Li) bp_check(L0, Li, SRO_M, plen, tlen, score_0, score_i, 0, bp_forward, bp);
    }
}

static int bialign_find_breakpoint(const view_t *w, const sro_penalties *pen,
                                   int cb, int ce, bp_t *bp, uint64_t *cells) {
    wfa_t F, R;
    view_t wr = *w; wr.rev = 1;
    const arena_mark mark = arena_get_mark();
    wfa_init(&F, w, pen, 1);
    wfa_init(&R, &wr, pen, 1);
    const int plen = w->plen, tlen = w->tlen;
    const int max_antidiagonal = plen + tlen - 1;
    const int scope = F.scope;
    const int gap_opening = F.two ? IMAX(F.o1, F.o2) : F.o1;
    long smax = 2L * ((long)F.o1 + (long)F.e1 * plen + (long)F.o1 + (long)F.e1 * tlen) + 1024;
    int score_f = 0, score_r = 0;
    bp->score = INT_MAX;
    wfa_level0(&F, cb);
    wfa_level0(&R, ce);
    int f_max_ak = level_max_ak(wfa_level(&F, 0));
    int r_max_ak = level_max_ak(wfa_level(&R, 0));
    int last_wf_forward = 0;
    int status = 0;
    for (;;) {
        if (f_max_ak + r_max_ak >= max_antidiagonal) break;
        ++score_f;
        wfa_step(&F, score_f);
        f_max_ak = level_max_ak(wfa_level(&F, score_f));
        last_wf_forward = 1;
        if (f_max_ak + r_max_ak >= max_antidiagonal) break;
        ++score_r;
        wfa_step(&R, score_r);
        r_max_ak = level_max_ak(wfa_level(&R, score_r));
        last_wf_forward = 0;
        if (score_f + score_r > smax) { status = -1; break; }
    }
    while (status == 0) {
        if (last_wf_forward) {
            const int min_score_r = (score_r > scope - 1) ? score_r - (scope - 1) : 0;
            if (score_f + min_score_r - gap_opening >= bp->score) break;
            bialign_overlap(&F, &R, score_f, score_r, 1, bp);
            ++score_r;
            wfa_step(&R, score_r);
        }
        const int min_score_f = (score_f > scope - 1) ? score_f - (scope - 1) : 0;
        if (min_score_f + score_r - gap_opening >= bp->score) break;
        bialign_overlap(&R, &F, score_r, score_f, 0, bp);
        ++score_f;
        wfa_step(&F, score_f);
        last_wf_forward = 1;
        if (score_f + score_r > smax) { status = -1; break; }
    }
    if (cells) *cells += F.cells + R.cells;
    wfa_free(&F);
    wfa_free(&R);
    arena_release(mark);
    if (status == 0 && bp->score == INT_MAX) status = -1;
    return status;
}

static int bialign(const uint8_t *p, int pb, int pe, const uint8_t *t, int tb,
                   int te, const sro_penalties *pen, int cb, int ce,
                   int score_remaining, cig_t *out, uint64_t *cells, int depth) {
    const int plen = pe - pb, tlen = te - tb;
    if (tlen == 0) { cig_push(out, 'D', plen); return 0; }
    if (plen == 0) { cig_push(out, 'I', tlen); return 0; }
    view_t w = { p + pb, t + tb, plen, tlen, 0 };
    if (score_remaining <= SRO_BIALIGN_FALLBACK_MIN_SCORE ||
        IMAX(plen, tlen) <= SRO_BIALIGN_FALLBACK_MIN_LENGTH || depth > 60) {
        int sc = wfa_full(&w, pen, cb, ce, out, cells);
        return sc < 0 ? sc : 0;
    }
    bp_t bp;
    int st = bialign_find_breakpoint(&w, pen, cb, ce, &bp, cells);
    if (st < 0) return st;
    const int bh = bp.off_f;
    const int bv = bp.off_f - bp.k_f;
    if (bv < 0 || bv > plen || bh < 0 || bh > tlen) return -3;
    st = bialign(p, pb, pb + bv, t, tb, tb + bh, pen, cb, bp.comp, bp.score_f,
                 out, cells, depth + 1);
    if (st < 0) return st;
    st = bialign(p, pb + bv, pe, t, tb + bh, te, pen, bp.comp, ce, bp.score_r,
                 out, cells, depth + 1);
    return st;
}

/* ---------------- public entry points ---------------- */
static int pen_valid(const sro_penalties *pen) {
    if (pen->match != 0) return 0;             /* only match == 0 restated */
    if (pen->mismatch <= 0 || pen->gap_open1 < 0 || pen->gap_ext1 <= 0) return 0;
    if (pen->gap_open2 >= 0 && pen->gap_ext2 <= 0) return 0;
    return 1;
}

int sro_cigar_score(const uint8_t *cigar, int n, const uint8_t *pattern,
                    int plen, const uint8_t *text, int tlen,
                    const sro_penalties *pen) {
    int v = 0, h = 0, score = 0, i = 0;
    const int two = pen->gap_open2 >= 0;
    while (i < n) {
        uint8_t op = cigar[i];
        int j = i;
        while (j < n && cigar[j] == op) j++;
        int len = j - i;
        if (op == 'M') {
            for (int q = 0; q < len; q++) {
                if (v >= plen || h >= tlen || pattern[v] != text[h]) return -1;
                v++; h++;
            }
        } else if (op == 'X') {
            for (int q = 0; q < len; q++) {
                if (v >= plen || h >= tlen || pattern[v] == text[h]) return -1;
                v++; h++;
            }
            score += len * pen->mismatch;
        } else if (op == 'I' || op == 'D') {
            int g = pen->gap_open1 + pen->gap_ext1 * len;
            if (two) g = IMIN(g, pen->gap_open2 + pen->gap_ext2 * len);
            score += g;
            if (op == 'I') h += len; else v += len;
        } else return -1;
        i = j;
    }
    if (v != plen || h != tlen) return -1;
    return score;
}

int sro_wfa_align(const uint8_t *pattern, int plen, const uint8_t *text,
                  int tlen, const sro_penalties *pen, int memory_mode,
                  uint8_t **cigar, int *cigar_len, int *score) {
    if (!pen_valid(pen) || plen < 0 || tlen < 0) return -1;
    cig_t out = {0, 0, 0};
    uint64_t cells = 0;
    int st;
    if (plen == 0 || tlen == 0) {
        cig_push(&out, 'D', plen);
        cig_push(&out, 'I', tlen);
        st = 0;
    } else if (memory_mode == SRO_MEM_ULTRALOW) {
        st = bialign(pattern, 0, plen, text, 0, tlen, pen, SRO_M, SRO_M,
                     INT_MAX, &out, &cells, 0);
    } else {
        view_t w = { pattern, text, plen, tlen, 0 };
        st = wfa_full(&w, pen, SRO_M, SRO_M, &out, &cells);
        if (st >= 0) st = 0;
    }
    g_last_cells = cells;
    if (st < 0) { free(out.b); return st; }
    if (score) *score = sro_cigar_score(out.b, out.n, pattern, plen, text, tlen, pen);
    if (cigar) { *cigar = out.b ? out.b : (uint8_t *)malloc(1); } else free(out.b);
    if (cigar_len) *cigar_len = out.n;
    return 0;
}

int sro_wfa_score(const uint8_t *pattern, int plen, const uint8_t *text,
                  int tlen, const sro_penalties *pen, int max_score,
                  int *score) {
    if (!pen_valid(pen) || plen <= 0 || tlen <= 0) return -1;
    view_t w = { pattern, text, plen, tlen, 0 };
    wfa_t a;
    const arena_mark mark = arena_get_mark();
    wfa_init(&a, &w, pen, 1);
    const int k_end = tlen - plen;
    long smax = (long)a.o1 + (long)a.e1 * plen + (long)a.o1 + (long)a.e1 * tlen + 64;
    wfa_level0(&a, SRO_M);
    int s = 0, res = INT_MAX;
    for (;;) {
        const level_t *L = wfa_level(&a, s);
        if (L && lv_get(L, SRO_M, k_end) >= tlen) { res = s; break; }
        if (s > smax) break;
        if (max_score >= 0 && s >= max_score) break;
        s++;
        wfa_step(&a, s);
    }
    g_last_cells = a.cells;
    wfa_free(&a);
    arena_release(mark);
    *score = res;
    return 0;
}

/* ---------------- independent Gotoh 2-piece affine DP ---------------- */
int sro_gotoh_score(const uint8_t *pattern, int plen, const uint8_t *text,
                    int tlen, const sro_penalties *pen) {
    const int INF = INT_MAX / 4;
    const int two = pen->gap_open2 >= 0;
    const int x = pen->mismatch, o1 = pen->gap_open1, e1 = pen->gap_ext1;
    const int o2 = two ? pen->gap_open2 : 0, e2 = two ? pen->gap_ext2 : 0;
    size_t n = (size_t)tlen + 1;
    int *M = (int *)malloc(sizeof(int) * n), *I1 = (int *)malloc(sizeof(int) * n),
        *I2 = (int *)malloc(sizeof(int) * n), *D1 = (int *)malloc(sizeof(int) * n),
        *D2 = (int *)malloc(sizeof(int) * n);
    /* row 0 */
    M[0] = 0; I1[0] = I2[0] = D1[0] = D2[0] = INF;
    for (int h = 1; h <= tlen; h++) {
        I1[h] = IMIN(M[h - 1] + o1 + e1, I1[h - 1] + e1);
        I2[h] = two ? IMIN(M[h - 1] + o2 + e2, I2[h - 1] + e2) : INF;
        D1[h] = D2[h] = INF;
        M[h] = IMIN(I1[h], I2[h]);
    }
    for (int v = 1; v <= plen; v++) {
        int diag = M[0];          /* M[v-1][0] */
        /* column 0 */
        int d1 = IMIN(M[0] + o1 + e1, D1[0] + e1);
        int d2 = two ? IMIN(M[0] + o2 + e2, D2[0] + e2) : INF;
        D1[0] = d1; D2[0] = d2; I1[0] = I2[0] = INF;
        M[0] = IMIN(d1, d2);
        for (int h = 1; h <= tlen; h++) {
            int up = M[h];        /* M[v-1][h] */
            int nd1 = IMIN(up + o1 + e1, D1[h] + e1);
            int nd2 = two ? IMIN(up + o2 + e2, D2[h] + e2) : INF;
            int ni1 = IMIN(M[h - 1] + o1 + e1, I1[h - 1] + e1);
            int ni2 = two ? IMIN(M[h - 1] + o2 + e2, I2[h - 1] + e2) : INF;
            int sub = diag + (pattern[v - 1] == text[h - 1] ? 0 : x);
            int m = IMIN(sub, IMIN(IMIN(nd1, nd2), IMIN(ni1, ni2)));
            diag = up;
            D1[h] = nd1; D2[h] = nd2; I1[h] = ni1; I2[h] = ni2; M[h] = m;
        }
    }
    int res = M[tlen];
    free(M); free(I1); free(I2); free(D1); free(D2);
    return res;
}

/* cigar_bytes_to_string: src/wfa.rs:9-38 */
char *sro_cigar_bytes_to_string(const uint8_t *cigar_bytes, int n) {
    size_t cap = 64, len = 0;
    char *s = (char *)malloc(cap);
    int i = 0;
    while (i < n) {
        uint8_t op = cigar_bytes[i];
        int count = 1, j = i + 1;
        while (j < n && cigar_bytes[j] == op) { count++; j++; }
        char op_char;
        switch (op) {
        case 'M': op_char = '='; break;
        case 'X': op_char = 'X'; break;
        case 'I': op_char = 'D'; break;   /* WFA2 'I' means standard 'D' */
        case 'D': op_char = 'I'; break;   /* WFA2 'D' means standard 'I' */
        default: op_char = '?'; break;
        }
        if (len + 24 > cap) { cap *= 2; s = (char *)realloc(s, cap); }
        len += (size_t)snprintf(s + len, cap - len, "%d%c", count, op_char);
        i = j;
    }
    s[len] = 0;
    return s;
}
