/* ORACLE (test infrastructure) -- Pos codec, uf_rush 0.2.1 and
 * BidirectedUnionFind restated in C11.  See sr_oracle.h for the rules. */
#include "sr_oracle.h"
#include <stdatomic.h>
#include <stdlib.h>
#include <stdio.h>

/* ---------------- Pos: src/pos.rs ---------------- */
sro_pos sro_make_pos(uint64_t offset, int is_reverse) {          /* :10-12 */
    return (offset << 1) | (uint64_t)(is_reverse ? 1 : 0);
}
int sro_is_rev(sro_pos p) { return (p & 1) == 1; }               /* :16-18 */
uint64_t sro_offset(sro_pos p) { return p >> 1; }                /* :22-24 */
sro_pos sro_incr_pos(sro_pos p) {                                /* :28-41 */
    if (sro_is_rev(p)) {
        uint64_t off = sro_offset(p);
        return off > 0 ? sro_make_pos(off - 1, 1) : p;
    }
    return sro_make_pos(sro_offset(p) + 1, 0);
}
sro_pos sro_decr_pos(sro_pos p) {                                /* :45-58 */
    if (sro_is_rev(p)) return sro_make_pos(sro_offset(p) + 1, 1);
    uint64_t off = sro_offset(p);
    return off > 0 ? sro_make_pos(off - 1, 0) : p;
}
sro_pos sro_flip_orientation(sro_pos p) { return p ^ 1; }        /* :62-64 */
char sro_orientation_char(sro_pos p) { return sro_is_rev(p) ? '-' : '+'; }
uint8_t sro_rc_base(uint8_t b) {                                 /* :78-87 */
    switch (b) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'N': case 'n': return 'N';
    default: return b;
    }
}

/* ---------------- uf_rush 0.2.1 src/lib.rs ---------------- */
#define RANK_BITS 6u                    /* lib.rs:4  usize::BITS.ilog2() */
#define PARENT_BITS 58u                 /* lib.rs:7 */
#define MAX_SIZE (UINT64_MAX >> RANK_BITS) /* lib.rs:10 */

struct sro_uf {
    _Atomic uint64_t *nodes;
    uint64_t size;
};

static inline uint64_t uf_encode(uint64_t parent, uint64_t rank) {  /* :228 */
    return parent | (rank << PARENT_BITS);
}
static inline uint64_t uf_parent(uint64_t n) { return n & MAX_SIZE; } /* :233 */
static inline uint64_t uf_rank(uint64_t n) { return n >> PARENT_BITS; } /* :238 */

sro_uf *sro_uf_new(uint64_t size) {                              /* :36-42 */
    if (size > MAX_SIZE) return NULL;
    sro_uf *u = (sro_uf *)malloc(sizeof(*u));
    u->size = size;
    u->nodes = (_Atomic uint64_t *)malloc(sizeof(uint64_t) * (size ? size : 1));
    for (uint64_t i = 0; i < size; i++) atomic_init(&u->nodes[i], i);
    return u;
}
void sro_uf_free(sro_uf *u) {
    if (!u) return;
    free((void *)u->nodes);
    free(u);
}
uint64_t sro_uf_size(const sro_uf *u) { return u->size; }
uint64_t *sro_uf_nodes(sro_uf *u) { return (uint64_t *)u->nodes; }

uint64_t sro_uf_find(sro_uf *u, uint64_t x) {                    /* :112-133 */
    if (x >= u->size) {   /* assert!(x < self.size()) :113 */
        fprintf(stderr, "sro_uf_find: index %llu out of bounds\n",
                (unsigned long long)x);
        abort();
    }
    uint64_t x_node = atomic_load_explicit(&u->nodes[x], memory_order_relaxed);
    while (x != uf_parent(x_node)) {
        uint64_t x_parent = uf_parent(x_node);
        uint64_t x_parent_node =
            atomic_load_explicit(&u->nodes[x_parent], memory_order_relaxed);
        uint64_t x_parent_parent = uf_parent(x_parent_node);
        uint64_t x_new_node = uf_encode(x_parent_parent, uf_rank(x_node));
        uint64_t expected = x_node;
        (void)atomic_compare_exchange_weak_explicit(
            &u->nodes[x], &expected, x_new_node, memory_order_release,
            memory_order_relaxed);
        x = x_parent_parent;
        x_node = atomic_load_explicit(&u->nodes[x], memory_order_relaxed);
    }
    return x;
}

int sro_uf_unite(sro_uf *u, uint64_t x, uint64_t y) {            /* :159-208 */
    for (;;) {
        uint64_t x_rep = sro_uf_find(u, x);
        uint64_t y_rep = sro_uf_find(u, y);
        if (x_rep == y_rep) return 0;
        uint64_t x_node =
            atomic_load_explicit(&u->nodes[x_rep], memory_order_relaxed);
        uint64_t y_node =
            atomic_load_explicit(&u->nodes[y_rep], memory_order_relaxed);
        uint64_t x_rank = uf_rank(x_node), y_rank = uf_rank(y_node);
        /* make x the smaller one: lower rank, or equal rank and smaller index
         * (:178-181) */
        if (x_rank > y_rank || (x_rank == y_rank && x_rep > y_rep)) {
            uint64_t t = x_rep; x_rep = y_rep; y_rep = t;
            t = x_rank; x_rank = y_rank; y_rank = t;
        }
        uint64_t cur_value = uf_encode(x_rep, x_rank);
        uint64_t new_value = uf_encode(y_rep, x_rank);
        if (atomic_compare_exchange_strong_explicit(
                &u->nodes[x_rep], &cur_value, new_value, memory_order_release,
                memory_order_acquire)) {
            if (x_rank == y_rank) {                               /* :194-203 */
                uint64_t cv = uf_encode(y_rep, y_rank);
                uint64_t nv = uf_encode(y_rep, y_rank + 1);
                (void)atomic_compare_exchange_weak_explicit(
                    &u->nodes[y_rep], &cv, nv, memory_order_release,
                    memory_order_relaxed);
            }
            return 1;
        }
    }
}

int sro_uf_same(sro_uf *u, uint64_t x, uint64_t y) {             /* :72-84 */
    for (;;) {
        uint64_t x_rep = sro_uf_find(u, x);
        uint64_t y_rep = sro_uf_find(u, y);
        if (x_rep == y_rep) return 1;
        uint64_t x_node =
            atomic_load_explicit(&u->nodes[x_rep], memory_order_relaxed);
        if (x_rep == uf_parent(x_node)) return 0;
    }
}

/* ------------- BidirectedUnionFind src/bidirected_union_find.rs -------- */
sro_uf *sro_buf_new(uint64_t max_offset) {                        /* :16-24 */
    return sro_uf_new((max_offset << 1) + 2);
}
sro_pos sro_buf_find(sro_uf *u, sro_pos p) { return sro_uf_find(u, p); }
void sro_buf_unite(sro_uf *u, sro_pos a, sro_pos b) {             /* :35-43 */
    if (a != b) sro_uf_unite(u, a, b);
}
int sro_buf_same(sro_uf *u, sro_pos a, sro_pos b) {               /* :46-54 */
    if (a == b) return 1;
    return sro_uf_find(u, a) == sro_uf_find(u, b);
}
void sro_buf_unite_matching_region(sro_uf *u, uint64_t seq1_offset,
    uint64_t seq2_offset, uint64_t seq1_local_start, uint64_t seq2_local_start,
    uint64_t match_length, int seq1_is_rc, uint64_t seq1_len) {   /* :60-98 */
    for (uint64_t i = 0; i < match_length; i++) {
        if (seq1_is_rc) {
            uint64_t rc_local_pos = seq1_local_start + i;
            uint64_t forward_local_pos = seq1_len - 1 - rc_local_pos;
            uint64_t seq1_global_offset = seq1_offset + forward_local_pos;
            sro_pos pos1_rev = sro_make_pos(seq1_global_offset, 1);
            sro_pos pos2_fwd =
                sro_make_pos(seq2_offset + seq2_local_start + i, 0);
            sro_buf_unite(u, pos1_rev, pos2_fwd);
        } else {
            sro_pos pos1_fwd =
                sro_make_pos(seq1_offset + seq1_local_start + i, 0);
            sro_pos pos2_fwd =
                sro_make_pos(seq2_offset + seq2_local_start + i, 0);
            sro_buf_unite(u, pos1_fwd, pos2_fwd);
        }
    }
}
void sro_buf_unite_matching_region_seq2_rc(sro_uf *u, uint64_t seq1_offset,
    uint64_t seq2_offset, uint64_t seq1_local_start, uint64_t seq2_local_start,
    uint64_t match_length, int seq2_is_rc, uint64_t seq2_len) {   /* :102-129 */
    for (uint64_t i = 0; i < match_length; i++) {
        if (seq2_is_rc) {
            sro_pos pos1_fwd =
                sro_make_pos(seq1_offset + seq1_local_start + i, 0);
            uint64_t seq2_rc_pos = seq2_len - 1 - (seq2_local_start + i);
            sro_pos pos2_rev = sro_make_pos(seq2_offset + seq2_rc_pos, 1);
            sro_buf_unite(u, pos1_fwd, pos2_rev);
        } else {
            sro_pos pos1_fwd =
                sro_make_pos(seq1_offset + seq1_local_start + i, 0);
            sro_pos pos2_fwd =
                sro_make_pos(seq2_offset + seq2_local_start + i, 0);
            sro_buf_unite(u, pos1_fwd, pos2_fwd);
        }
    }
}
