/* ORACLE (test infrastructure) -- linear-chain compaction + sequential renumbering of the induced graph,
 * restating src/bidirected_ops.rs:
 *   compact()                         :91-112
 *   find_simple_components()          :115-275   (adjacency incl. implied reverse edges :126-143,
 *                                                 are_perfect_neighbors :146-207, greedy forward chains :210-272)
 *   merge_component_v2()              :279-490   (handle mapping :307-315, path validation :318-363, path rewrite
 *                                                 :369-413, edge rewrite :416-477, node removal :480-487)
 *   renumber_nodes_sequentially()     :75-89 + apply_node_id_mapping :21-70
 *   reverse_complement of node text   src/bidirected_graph.rs:73-85 (N/n -> N)
 * as driven by write_bidirected_gfa (src/bidirected_gfa_writer.rs:39-51: compact(); renumber) for --no-sort
 * without --no-compact.  Deliberately literal (every predicate rescans the paths like the reference does), so it is
 * quadratic: for test-sized graphs only.  The reference keeps edges in a HashSet; nothing below depends on their
 * iteration order (adjacency lists are only used when they hold exactly one element), and the GFA is compared on
 * its canonical form (L lines as a sorted multiset).
 *
 * Interface: GFA text (as sro_build_gfa writes it for --no-compact) in, compacted GFA text out.
 */
#include "sr_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t hnd;                       /* Handle: node_id << 1 | is_reverse (bidirected_graph.rs:9-64) */
#define H_ID(h) ((h) >> 1)
#define H_FLIP(h) ((h) ^ 1ULL)

typedef struct { uint8_t *seq; uint64_t len; int alive; } cnode;
typedef struct { char *name; hnd *steps; uint64_t n; } cpath;
typedef struct { hnd from, to; } cedge;
typedef struct {
    cnode *nodes; uint64_t nnodes;          /* nodes.len(): slot 0 is never used */
    cedge *edges; uint64_t nedges, ecap;
    cpath *paths; uint64_t npaths;
} cgraph;

static uint8_t rc_node_base(uint8_t b) {    /* bidirected_graph.rs:73-85 */
    switch (b) {
    case 'A': case 'a': return 'T'; case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C';
    case 'N': case 'n': return 'N';
    default: return b;
    }
}

static void g_add_node(cgraph *g, uint64_t id, uint8_t *seq, uint64_t len) {    /* add_node ops:804-810 */
    if (id >= g->nnodes) {
        g->nodes = (cnode *)realloc(g->nodes, sizeof(cnode) * (id + 1));
        for (uint64_t i = g->nnodes; i <= id; i++) { g->nodes[i].seq = NULL; g->nodes[i].len = 0; g->nodes[i].alive = 0; }
        g->nnodes = id + 1;
    }
    free(g->nodes[id].seq);
    g->nodes[id].seq = seq; g->nodes[id].len = len; g->nodes[id].alive = 1;
}
static int g_has_edge(const cedge *e, uint64_t n, hnd f, hnd t) {
    for (uint64_t i = 0; i < n; i++) if (e[i].from == f && e[i].to == t) return 1;
    return 0;
}
static void eset_push(cedge **e, uint64_t *n, uint64_t *cap, hnd f, hnd t) {     /* insert of a key known to be new */
    if (*n == *cap) { *cap = *cap ? *cap * 2 : 256; *e = (cedge *)realloc(*e, sizeof(cedge) * *cap); }
    (*e)[*n].from = f; (*e)[*n].to = t; (*n)++;
}
static void eset_insert(cedge **e, uint64_t *n, uint64_t *cap, hnd f, hnd t) {   /* HashSet::insert */
    if (g_has_edge(*e, *n, f, t)) return;
    eset_push(e, n, cap, f, t);
}

/* adjacency of one handle, ops:126-143: forward[from] += to, backward[to] += from, and for the implied reverse edge
 * forward[to.flip()] += from.flip(), backward[from.flip()] += to.flip() */
static uint64_t adj_forward(const cgraph *g, hnd h, hnd *only) {
    uint64_t c = 0;
    for (uint64_t i = 0; i < g->nedges; i++) {
        if (g->edges[i].from == h) { c++; *only = g->edges[i].to; }
        if (H_FLIP(g->edges[i].to) == h) { c++; *only = H_FLIP(g->edges[i].from); }
    }
    return c;
}
static uint64_t adj_backward(const cgraph *g, hnd h) {
    uint64_t c = 0;
    for (uint64_t i = 0; i < g->nedges; i++) {
        if (g->edges[i].to == h) c++;
        if (H_FLIP(g->edges[i].from) == h) c++;
    }
    return c;
}

static int are_perfect_neighbors(const cgraph *g, hnd from, hnd to) {            /* ops:146-207 */
    for (uint64_t p = 0; p < g->npaths; p++) {
        const cpath *path = &g->paths[p];
        uint64_t from_to = 0, from_visits = 0;
        for (uint64_t i = 0; i < path->n; i++) {
            if (path->steps[i] == from) {
                from_visits++;
                if (i + 1 < path->n) { if (path->steps[i + 1] == to) from_to++; else return 0; }
                else return 0;
            }
        }
        if (from_visits > 0 && from_visits != from_to) return 0;
        const hnd from_rev = H_FLIP(from), to_rev = H_FLIP(to);
        uint64_t tr_visits = 0, tr_to_fr = 0;
        for (uint64_t i = 0; i < path->n; i++) {
            if (path->steps[i] == to_rev) {
                tr_visits++;
                if (i + 1 < path->n) { if (path->steps[i + 1] == from_rev) tr_to_fr++; else return 0; }
                else return 0;
            }
        }
        if (tr_visits > 0 && tr_visits != tr_to_fr) return 0;
    }
    return 1;
}

typedef struct { hnd *h; uint64_t n; } chain_t;

static chain_t *find_simple_components(const cgraph *g, uint64_t *ncomp) {        /* ops:115-275 */
    const uint64_t nh = g->nnodes * 2;
    uint8_t *visited = (uint8_t *)calloc(nh ? nh : 1, 1), *merged = (uint8_t *)calloc(g->nnodes ? g->nnodes : 1, 1);
    chain_t *comps = NULL; uint64_t nc = 0, ccap = 0;
    for (uint64_t id = 0; id < g->nnodes; id++) {
        if (!g->nodes[id].alive) continue;
        for (int rev = 0; rev < 2; rev++) {
            const hnd handle = (id << 1) | (hnd)rev;
            if (visited[handle]) continue;
            hnd only = 0;
            if (adj_forward(g, handle, &only) != 1) continue;              /* out_degree == 1 (:218-221) */
            hnd *chain = (hnd *)malloc(sizeof(hnd) * (nh ? nh : 1));
            uint64_t n = 0;
            chain[n++] = handle; visited[handle] = 1;
            hnd current = handle;
            for (;;) {                                                     /* :227-253 */
                hnd next = 0;
                const uint64_t outd = adj_forward(g, current, &next);
                if (outd == 0) break;                                      /* `while let Some(nexts)`: no entry */
                if (outd != 1) break;
                if (adj_backward(g, next) != 1 || visited[next]) break;
                if (!are_perfect_neighbors(g, current, next)) break;
                chain[n++] = next; visited[next] = 1; current = next;
                hnd dummy = 0;
                if (adj_forward(g, next, &dummy) != 1) break;
            }
            if (n >= 2) {                                                  /* :255-270 */
                int already = 0;
                for (uint64_t i = 0; i < n; i++) if (merged[H_ID(chain[i])]) { already = 1; break; }
                if (!already) {
                    for (uint64_t i = 0; i < n; i++) merged[H_ID(chain[i])] = 1;
                    if (nc == ccap) { ccap = ccap ? ccap * 2 : 64; comps = (chain_t *)realloc(comps, sizeof(chain_t) * ccap); }
                    comps[nc].h = chain; comps[nc].n = n; nc++;
                    continue;
                }
            }
            free(chain);
        }
    }
    free(visited); free(merged);
    *ncomp = nc;
    return comps;
}

/* handle_mapping: HashMap insert order of ops:307-315 (a later insert of the same key wins) */
static int map_lookup(const hnd *handles, uint64_t n, hnd key, uint64_t *chain_pos) {
    int found = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (handles[i] == key) { *chain_pos = i; found = 1; }
        if (H_FLIP(handles[i]) == key) { *chain_pos = n - 1 - i; found = 1; }
    }
    return found;
}

static int merge_component_v2(cgraph *g, const hnd *handles, uint64_t n) {        /* ops:279-490 */
    if (n < 2) return 0;
    uint64_t new_len = 0;
    for (uint64_t i = 0; i < n; i++) if (H_ID(handles[i]) < g->nnodes && g->nodes[H_ID(handles[i])].alive) new_len += g->nodes[H_ID(handles[i])].len;
    uint8_t *new_seq = (uint8_t *)malloc(new_len ? new_len : 1);
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; i++) {                                      /* :290-298 */
        const uint64_t id = H_ID(handles[i]);
        if (id >= g->nnodes || !g->nodes[id].alive) continue;
        const cnode *nd = &g->nodes[id];
        if (handles[i] & 1) for (uint64_t k = 0; k < nd->len; k++) new_seq[w++] = rc_node_base(nd->seq[nd->len - 1 - k]);
        else { memcpy(new_seq + w, nd->seq, nd->len); w += nd->len; }
    }
    const uint64_t new_id = g->nnodes;                                      /* next_node_id ops:692-694 */
    const hnd new_fwd = new_id << 1, new_rev = (new_id << 1) | 1;
    hnd *rev_chain = (hnd *)malloc(sizeof(hnd) * n);
    for (uint64_t i = 0; i < n; i++) rev_chain[i] = H_FLIP(handles[n - 1 - i]);
    /* validation :318-363 */
    for (uint64_t p = 0; p < g->npaths; p++) {
        const cpath *path = &g->paths[p];
        uint64_t i = 0;
        while (i < path->n) {
            uint64_t chain_pos = 0;
            if (map_lookup(handles, n, path->steps[i], &chain_pos)) {
                if (chain_pos == 0) {
                    if (i + n <= path->n) {
                        int complete = 1;
                        for (uint64_t j = 0; j < n; j++) if (path->steps[i + j] != handles[j]) { complete = 0; break; }
                        if (complete) { i += n; continue; }
                    }
                }
                if (path->steps[i] == rev_chain[0]) {
                    if (i + n <= path->n) {
                        int complete = 1;
                        for (uint64_t j = 0; j < n; j++) if (path->steps[i + j] != rev_chain[j]) { complete = 0; break; }
                        if (complete) { i += n; continue; }
                    }
                }
                free(new_seq); free(rev_chain);
                return 0;
            }
            i++;
        }
    }
    g_add_node(g, new_id, new_seq, new_len);
    /* path rewrite :369-413 */
    for (uint64_t p = 0; p < g->npaths; p++) {
        cpath *path = &g->paths[p];
        hnd *ns = (hnd *)malloc(sizeof(hnd) * (path->n ? path->n : 1));
        uint64_t m = 0, i = 0;
        while (i < path->n) {
            if (i + n <= path->n) {
                int fwd = 1;
                for (uint64_t j = 0; j < n; j++) if (path->steps[i + j] != handles[j]) { fwd = 0; break; }
                if (fwd) { ns[m++] = new_fwd; i += n; continue; }
            }
            if (i + n <= path->n) {
                int rv = 1;
                for (uint64_t j = 0; j < n; j++) if (path->steps[i + j] != rev_chain[j]) { rv = 0; break; }
                if (rv) { ns[m++] = new_rev; i += n; continue; }
            }
            ns[m++] = path->steps[i]; i++;
        }
        free(path->steps); path->steps = ns; path->n = m;
    }
    /* edge rewrite :416-477 */
    const hnd first = handles[0], last = handles[n - 1];
    cedge *ne = NULL; uint64_t nn = 0, ncap = 0;
    for (uint64_t e = 0; e < g->nedges; e++) {
        const cedge ed = g->edges[e];
        int from_in = 0, to_in = 0;
        for (uint64_t i = 0; i < n; i++) { if (H_ID(handles[i]) == H_ID(ed.from)) from_in = 1; if (H_ID(handles[i]) == H_ID(ed.to)) to_in = 1; }
        if (from_in && to_in) continue;
        else if (!from_in && !to_in) eset_push(&ne, &nn, &ncap, ed.from, ed.to);   /* distinct before, untouched: still distinct from each other */
        else if (from_in && !to_in) {
            if (ed.from == last) eset_insert(&ne, &nn, &ncap, new_fwd, ed.to);
            if (ed.from == H_FLIP(first)) eset_insert(&ne, &nn, &ncap, new_rev, ed.to);
        } else {
            if (ed.to == first) eset_insert(&ne, &nn, &ncap, ed.from, new_fwd);
            if (ed.to == H_FLIP(last)) eset_insert(&ne, &nn, &ncap, ed.from, new_rev);
        }
    }
    free(g->edges); g->edges = ne; g->nedges = nn; g->ecap = ncap;
    for (uint64_t i = 0; i < n; i++) {                                      /* :480-487 */
        const uint64_t id = H_ID(handles[i]);
        if (id < g->nnodes) { g->nodes[id].alive = 0; free(g->nodes[id].seq); g->nodes[id].seq = NULL; g->nodes[id].len = 0; }
    }
    free(rev_chain);
    return 1;
}

static void compact(cgraph *g) {                                                  /* ops:91-112 */
    int compacted = 1;
    while (compacted) {
        compacted = 0;
        uint64_t nc = 0;
        chain_t *comps = find_simple_components(g, &nc);
        for (uint64_t c = 0; c < nc; c++) {
            if (comps[c].n >= 2 && merge_component_v2(g, comps[c].h, comps[c].n)) compacted = 1;
            free(comps[c].h);
        }
        free(comps);
    }
}

static void renumber_nodes_sequentially(cgraph *g) {                              /* ops:75-89, 21-70 */
    uint64_t *map = (uint64_t *)calloc(g->nnodes ? g->nnodes : 1, sizeof(uint64_t));
    uint64_t new_id = 1;
    for (uint64_t id = 0; id < g->nnodes; id++) if (g->nodes[id].alive) map[id] = new_id++;
    cnode *nn = (cnode *)calloc(new_id, sizeof(cnode));
    for (uint64_t id = 0; id < g->nnodes; id++) if (g->nodes[id].alive) nn[map[id]] = g->nodes[id];
    free(g->nodes); g->nodes = nn; g->nnodes = new_id;
    cedge *ne = NULL; uint64_t n2 = 0, cap = 0;
    for (uint64_t e = 0; e < g->nedges; e++)
        eset_push(&ne, &n2, &cap, (map[H_ID(g->edges[e].from)] << 1) | (g->edges[e].from & 1),     /* the id map is injective */
                  (map[H_ID(g->edges[e].to)] << 1) | (g->edges[e].to & 1));
    free(g->edges); g->edges = ne; g->nedges = n2; g->ecap = cap;
    for (uint64_t p = 0; p < g->npaths; p++)
        for (uint64_t i = 0; i < g->paths[p].n; i++)
            g->paths[p].steps[i] = (map[H_ID(g->paths[p].steps[i])] << 1) | (g->paths[p].steps[i] & 1);
    free(map);
}

/* ---- GFA text <-> cgraph (the subset sro_build_gfa writes: H, S, L ... 0M, P ... *) ---- */
static const char *next_field(const char *p, const char *end, const char **fs, size_t *fl) {
    *fs = p;
    while (p < end && *p != '\t' && *p != '\n') p++;
    *fl = (size_t)(p - *fs);
    return p;
}

/* Handle codec, src/bidirected_graph.rs:9-64: node id << 1 | reverse bit, Display "{id}{+|-}" */
uint64_t sro_handle_new(uint64_t node_id, int is_reverse) { return (node_id << 1) | (is_reverse ? 1u : 0u); }
uint64_t sro_handle_node_id(uint64_t h) { return H_ID(h); }
int sro_handle_is_reverse(uint64_t h) { return (int)(h & 1); }
char sro_handle_orientation_char(uint64_t h) { return (h & 1) ? '-' : '+'; }
uint64_t sro_handle_flip(uint64_t h) { return H_FLIP(h); }

static char *gfa_text_pass(const char *gfa, int do_compact, uint64_t *n_nodes, uint64_t *n_edges) {
    cgraph g; memset(&g, 0, sizeof(g));
    const char *p = gfa, *end = gfa + strlen(gfa);
    uint64_t pcap = 0;
    while (p < end) {
        const char *ls = p;
        const char *le = memchr(p, '\n', (size_t)(end - p));
        if (!le) le = end;
        if (le > ls + 1 && ls[1] == '\t') {
            const char *f; size_t fl;
            const char *q = ls + 2;
            if (ls[0] == 'S') {
                q = next_field(q, le, &f, &fl);
                uint64_t id = strtoull(f, NULL, 10);
                q = next_field(q + 1, le, &f, &fl);
                uint8_t *s = (uint8_t *)malloc(fl ? fl : 1); memcpy(s, f, fl);
                g_add_node(&g, id, s, fl);
            } else if (ls[0] == 'L') {
                q = next_field(q, le, &f, &fl); uint64_t a = strtoull(f, NULL, 10);
                q = next_field(q + 1, le, &f, &fl); int ar = f[0] == '-';
                q = next_field(q + 1, le, &f, &fl); uint64_t b = strtoull(f, NULL, 10);
                q = next_field(q + 1, le, &f, &fl); int br = f[0] == '-';
                eset_insert(&g.edges, &g.nedges, &g.ecap, (a << 1) | (hnd)ar, (b << 1) | (hnd)br);
            } else if (ls[0] == 'P') {
                q = next_field(q, le, &f, &fl);
                if (g.npaths == pcap) { pcap = pcap ? pcap * 2 : 16; g.paths = (cpath *)realloc(g.paths, sizeof(cpath) * pcap); }
                cpath *pa = &g.paths[g.npaths++];
                pa->name = (char *)malloc(fl + 1); memcpy(pa->name, f, fl); pa->name[fl] = 0;
                q = next_field(q + 1, le, &f, &fl);
                uint64_t cnt = fl ? 1 : 0;
                for (size_t i = 0; i < fl; i++) if (f[i] == ',') cnt++;
                pa->steps = (hnd *)malloc(sizeof(hnd) * (cnt ? cnt : 1)); pa->n = 0;
                const char *s = f, *se = f + fl;
                while (s < se) {
                    uint64_t id = 0;
                    while (s < se && *s >= '0' && *s <= '9') { id = id * 10 + (uint64_t)(*s - '0'); s++; }
                    int r = (s < se && *s == '-');
                    if (s < se) s++;
                    if (s < se && *s == ',') s++;
                    pa->steps[pa->n++] = (id << 1) | (hnd)r;
                }
            }
        }
        p = le < end ? le + 1 : end;
    }
    if (do_compact) { compact(&g); renumber_nodes_sequentially(&g); }
    /* write_gfa ops:880-925 */
    size_t cap = 1 << 16, len = 0;
    char *out = (char *)malloc(cap);
#define ENSURE(n) do { while (len + (n) + 64 > cap) { cap *= 2; out = (char *)realloc(out, cap); } } while (0)
    ENSURE(16); len += (size_t)sprintf(out + len, "H\tVN:Z:1.0\n");
    uint64_t live = 0;
    for (uint64_t id = 0; id < g.nnodes; id++) {
        if (!g.nodes[id].alive) continue;
        live++;
        ENSURE(g.nodes[id].len + 32);
        len += (size_t)sprintf(out + len, "S\t%llu\t", (unsigned long long)id);
        memcpy(out + len, g.nodes[id].seq, g.nodes[id].len); len += g.nodes[id].len;
        out[len++] = '\n';
    }
    for (uint64_t e = 0; e < g.nedges; e++) {
        ENSURE(64);
        len += (size_t)sprintf(out + len, "L\t%llu\t%c\t%llu\t%c\t0M\n", (unsigned long long)H_ID(g.edges[e].from),
                               (g.edges[e].from & 1) ? '-' : '+', (unsigned long long)H_ID(g.edges[e].to), (g.edges[e].to & 1) ? '-' : '+');
    }
    for (uint64_t pi = 0; pi < g.npaths; pi++) {
        ENSURE(strlen(g.paths[pi].name) + 8);
        len += (size_t)sprintf(out + len, "P\t%s\t", g.paths[pi].name);
        for (uint64_t i = 0; i < g.paths[pi].n; i++) {
            ENSURE(32);
            len += (size_t)sprintf(out + len, "%s%llu%c", i ? "," : "", (unsigned long long)H_ID(g.paths[pi].steps[i]),
                                   (g.paths[pi].steps[i] & 1) ? '-' : '+');
        }
        ENSURE(8); len += (size_t)sprintf(out + len, "\t*\n");
    }
    out[len] = 0;
#undef ENSURE
    if (n_nodes) *n_nodes = live;
    if (n_edges) *n_edges = g.nedges;
    for (uint64_t id = 0; id < g.nnodes; id++) free(g.nodes[id].seq);
    free(g.nodes); free(g.edges);
    for (uint64_t pi = 0; pi < g.npaths; pi++) { free(g.paths[pi].name); free(g.paths[pi].steps); }
    free(g.paths);
    return out;
}

char *sro_compact_gfa(const char *gfa, uint64_t *n_nodes, uint64_t *n_edges) { return gfa_text_pass(gfa, 1, n_nodes, n_edges); }
/* the graph of a GFA text written back by write_gfa (ops:880-925) unchanged: the writer's line formats on their own */
char *sro_rewrite_gfa(const char *gfa, uint64_t *n_nodes, uint64_t *n_edges) { return gfa_text_pass(gfa, 0, n_nodes, n_edges); }
/* BiPath::get_sequence (src/bidirected_graph.rs:113-154): spell path `index` of a GFA text; malloc'd, NUL-terminated */
char *sro_gfa_path_sequence(const char *gfa, uint64_t index) {
    /* nodes by id from the S lines, then the steps of the index-th P line */
    const char *p = gfa, *end = gfa + strlen(gfa);
    uint64_t cap = 16, nn = 0, seen = 0;
    struct nd { uint64_t id; const char *s; size_t n; } *nodes = malloc(sizeof(*nodes) * cap);
    const char *steps = NULL; size_t steps_n = 0;
    while (p < end) {
        const char *le = memchr(p, '\n', (size_t)(end - p));
        if (!le) le = end;
        const char *f; size_t fl;
        if (le > p + 1 && p[0] == 'S' && p[1] == '\t') {
            const char *q = next_field(p + 2, le, &f, &fl);
            if (nn == cap) { cap *= 2; nodes = realloc(nodes, sizeof(*nodes) * cap); }
            nodes[nn].id = strtoull(f, NULL, 10);
            next_field(q + 1, le, &f, &fl);
            nodes[nn].s = f; nodes[nn].n = fl; nn++;
        } else if (le > p + 1 && p[0] == 'P' && p[1] == '\t') {
            if (seen++ == index) { const char *q = next_field(p + 2, le, &f, &fl); next_field(q + 1, le, &steps, &steps_n); }
        }
        p = le < end ? le + 1 : end;
    }
    size_t ocap = 64, on = 0;
    char *out = malloc(ocap);
    const char *s = steps, *se = steps ? steps + steps_n : NULL;
    while (s && s < se) {
        uint64_t id = 0;
        while (s < se && *s >= '0' && *s <= '9') { id = id * 10 + (uint64_t)(*s - '0'); s++; }
        const int r = (s < se && *s == '-');
        if (s < se) s++;
        if (s < se && *s == ',') s++;
        for (uint64_t i = 0; i < nn; i++) if (nodes[i].id == id) {
            while (on + nodes[i].n + 1 > ocap) { ocap *= 2; out = realloc(out, ocap); }
            for (size_t k = 0; k < nodes[i].n; k++)
                out[on++] = r ? (char)rc_node_base((uint8_t)nodes[i].s[nodes[i].n - 1 - k]) : nodes[i].s[k];
            break;
        }
    }
    out[on] = 0;
    free(nodes);
    return out;
}
