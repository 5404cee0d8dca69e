// sr_host.cpp -- host side of the C ABI declared in include/seqrush_amd.h.
// Packs sequences (2-bit, forward + reverse complement), builds the ordered
// pair list, owns device memory / stream / events and launches the kernels of
// sr_device.hip.  No CPU alignment path exists here: without a HIP device the
// compute entry points fail with SR_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <climits>
#include <functional>
#include <queue>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "../../include/seqrush_amd.h"
#include "sr_internal.h"
#include "sr_graph.h"

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
extern "C" const char *sr_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------ environment knobs
// Every environment variable the load path reads, in ONE table: name, what unset means, what it does.  knob() is the only
// place that calls getenv for them and refuses a name that is not listed; sr_ctx_workspace_report() dumps every knob that is
// set ("knobs": {...}), so a measurement carries every non-default setting it ran with.
struct SrKnob { const char *name, *unset, *doc; };
static const SrKnob SR_KNOBS[] = {
    {"SR_ALIGN_IMPL", "2 when the penalties have a blocked instance", "1 = level-per-pass kernel (sr_align_bfs_kernel), 2 = score-blocked kernel"},
    {"SR_BLK_LEVELS", "10 for x=5, o1+e1=10", "5 = generic 5-level blocked instance instead of the exact 10-level one"},
    {"SR_ALIGN_THREADS", "by pairs per CU", "threads per workgroup: 64 | 128 | 256 | 512 | 1024"},
    {"SR_WG_PER_CU", "4", "workgroups per CU the launch is sized for"},
    {"SR_NWG", "CUs x workgroups per CU", "cap on workgroups (several pairs per workgroup on small inputs)"},
    {"SR_STATIC_LDS_KB", "23-33 / 28 / 6", "static LDS of the kernel assumed when sizing workgroups per CU (A/B builds with other tables)"},
    {"SR_RING_U16", "1 below 57 k", "0 = 32-bit searches keep 32-bit ring rows"},
    {"SR_LAZY_ID", "1", "0 = searches store their I/D rows from the first level (no recompute pass)"},
    {"SR_HIST_JOBS", "4", "worst-case base cases the per-workgroup history holds (1..16)"},
    {"SR_BFS_BASE_JOBS", "16 (lean builds 4 / 8)", "base cases per batch"},
    {"SR_CIGAR_ARENA_OPS", "25 % of free memory", "CIGAR arena in ops (tests: force batches)"},
    {"SR_POISON_ROWS", "off", "fill the row workspaces with this offset before the run (stale-row tests)"},
    {"SR_PREORIENT", "by pairs per CU and LDS", "1 / 0 = orientation as its own kernel / inside the alignment kernel"},
    {"SR_NO_REORDER", "off", "keep list order instead of the cost-sorted dequeue order"},
    {"SR_ORIENT_LEVELS", "off", "orientation level by level even for the default penalties"},
    {"SR_NO_KBITS", "off", "no q-gram bound in the orientation kernel"},
    {"SR_NO_FUSED_UNITE", "off", "sr_ctx_run launches sr_unite_kernel per batch instead of uniting inside the blocked alignment kernel"},
    {"SR_PROFILE_TICKS", "0", "1 = launch the instrumented instance (100 MHz tick counters)"},
    {"SR_FORCE_INT32", "off", "tests: 32-bit offsets (and the 16-bit ring of 32-bit searches) whatever the sequence length"},
    {"SR_TEST_BASE_LEVELS", "off", "tests: cap on the levels a base case is given at first (forces the re-queue path)"},
    {"SR_BOUNDS_DETAIL", "off", "1 = (bounds-checked build) keep the first offending access in counters[40..43]"},
};
static const char *knob(const char *name) {
    for (const SrKnob &k : SR_KNOBS)
        if (!strcmp(k.name, name)) return getenv(name);
    fprintf(stderr, "seqrush_amd: internal error: environment knob %s is not in SR_KNOBS\n", name);
    abort();
}
static std::string knobs_json() {
    std::string out = "{";
    for (const SrKnob &k : SR_KNOBS)
        if (const char *v = getenv(k.name)) {
            if (out.size() > 1) out += ", ";
            out += std::string("\"") + k.name + "\": \"";
            for (const char *q = v; *q; q++) if (*q != '"' && *q != '\\' && (unsigned char)*q >= 32) out += *q;
            out += "\"";
        }
    return out + "}";
}
// (name, unset, doc) lines of the table: `seqrush_mi355x --knobs`, scripts
extern "C" const char *sr_knobs_doc(void) {
    static std::string doc;
    if (doc.empty())
        for (const SrKnob &k : SR_KNOBS) doc += std::string(k.name) + "\t" + k.unset + "\t" + k.doc + "\n";
    return doc.c_str();
}
extern "C" int sr_abi_version(void) { return SR_ABI_VERSION; }
extern "C" void sr_free(void *p) { free(p); }
extern "C" int sr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

#define HIPCHK(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            return fail(SR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

// ------------------------------------------------------------------ params
extern "C" void sr_default_params(sr_params *p) {
    memset(p, 0, sizeof(*p));
    p->match_score = 0; p->mismatch_penalty = 5; p->gap_open1 = 8; p->gap_ext1 = 2;
    p->gap_open2 = 24; p->gap_ext2 = 1;                         // seqrush.rs:45
    p->ori_match = 0; p->ori_mismatch = 1; p->ori_gap_open = 1; p->ori_gap_ext = 1;  // :49
    p->min_match_len = 0; p->max_divergence = -1.0; p->exclude_self = 0;
    p->memory_mode = SR_MEM_ULTRALOW; p->sparsify_kind = SR_SPARSE_NONE; p->sparsify_factor = 1.0;
    p->sparsify_seed = 42; p->canonical_labels = 0; p->device = 0;
    p->shard_rank = 0; p->shard_count = 1;
    p->tree_k_nearest = 0; p->tree_k_farthest = 0; p->tree_rand_frac = 0.0; p->tree_kmer = 16;
}

static bool parse_i32(const std::string &s, int32_t *out) {     // Rust str::parse::<i32>
    if (s.empty() || s.size() > 15) return false;
    size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
    if (i == s.size()) return false;
    for (size_t j = i; j < s.size(); j++) if (s[j] < '0' || s[j] > '9') return false;
    long long v = atoll(s.c_str());
    if (v > INT32_MAX || v < INT32_MIN) return false;
    *out = (int32_t)v;
    return true;
}
static std::vector<std::string> split(const std::string &s, char c) {
    std::vector<std::string> out;
    size_t st = 0;
    for (;;) {
        size_t p = s.find(c, st);
        if (p == std::string::npos) { out.push_back(s.substr(st)); break; }
        out.push_back(s.substr(st, p - st));
        st = p + 1;
    }
    return out;
}

extern "C" int sr_parse_scores(const char *s, sr_params *p) {   // seqrush.rs:165-217
    auto parts = split(s, ',');
    if (parts.size() < 4)
        return fail(SR_ERR_INVALID, "Scores must have at least 4 values: match,mismatch,gap1_open,gap1_extend");
    if (parts.size() > 6) return fail(SR_ERR_INVALID, "Too many score values provided (max 6)");
    int32_t v[6] = {0, 0, 0, 0, -1, -1};
    static const char *nm[6] = {"match score", "mismatch penalty", "gap1_open penalty",
                                "gap1_extend penalty", "gap2_open penalty", "gap2_extend penalty"};
    for (int i = 0; i < 4; i++)
        if (!parse_i32(parts[i], &v[i])) return fail(SR_ERR_INVALID, std::string("Invalid ") + nm[i] + ": " + parts[i]);
    if (parts.size() >= 6)
        for (int i = 4; i < 6; i++)
            if (!parse_i32(parts[i], &v[i])) return fail(SR_ERR_INVALID, std::string("Invalid ") + nm[i] + ": " + parts[i]);
    p->match_score = v[0]; p->mismatch_penalty = v[1]; p->gap_open1 = v[2]; p->gap_ext1 = v[3];
    p->gap_open2 = v[4]; p->gap_ext2 = v[5];
    return SR_OK;
}

extern "C" int sr_parse_orientation_scores(const char *s, sr_params *p) {   // :219-250
    auto parts = split(s, ',');
    if (parts.size() != 4)
        return fail(SR_ERR_INVALID, "Orientation scores must have exactly 4 values: match,mismatch,gap_open,gap_extend");
    int32_t v[4];
    for (int i = 0; i < 4; i++)
        if (!parse_i32(parts[i], &v[i])) return fail(SR_ERR_INVALID, "Invalid orientation score: " + parts[i]);
    p->ori_match = v[0]; p->ori_mismatch = v[1]; p->ori_gap_open = v[2]; p->ori_gap_ext = v[3];
    return SR_OK;
}

static bool parse_f64(const std::string &s, double *out) {
    if (s.empty() || isspace((unsigned char)s[0])) return false;
    for (char ch : s) if (ch == 'x' || ch == 'X') return false;     // strtod takes hex floats, Rust's f64::from_str does not
    char *end;
    double v = strtod(s.c_str(), &end);
    if (*end != 0) return false;
    *out = v;
    return true;
}

extern "C" int sr_parse_sparsification(const char *cs, sr_params *p) {      // :356-431
    std::string s(cs);
    if (s == "none" || s == "1.0") { p->sparsify_kind = SR_SPARSE_NONE; return SR_OK; }
    if (s == "auto") { p->sparsify_kind = SR_SPARSE_AUTO; return SR_OK; }
    double f;
    if (s.rfind("random:", 0) == 0) {
        if (!parse_f64(s.substr(7), &f)) return fail(SR_ERR_INVALID, "Invalid random factor: " + s);
        if (f > 0.0 && f <= 1.0) { p->sparsify_kind = SR_SPARSE_RANDOM; p->sparsify_factor = f; return SR_OK; }
        return fail(SR_ERR_INVALID, "Random factor must be in (0.0, 1.0]");
    }
    if (s.rfind("connectivity:", 0) == 0) {
        if (!parse_f64(s.substr(13), &f)) return fail(SR_ERR_INVALID, "Invalid connectivity probability: " + s);
        if (f > 0.0 && f <= 1.0) { p->sparsify_kind = SR_SPARSE_CONNECTIVITY; p->sparsify_factor = f; return SR_OK; }
        return fail(SR_ERR_INVALID, "Connectivity probability must be in (0.0, 1.0]");
    }
    if (s.rfind("tree:", 0) == 0) {                              // tree:neighbor[,stranger[,random[,k-mer]]] (:378-418)
        auto parts = split(s.substr(5), ',');
        if (parts.empty() || parts.size() > 4)
            return fail(SR_ERR_INVALID, "Tree sampling requires 1-4 values: tree:neighbor[,stranger[,random[,k-mer]]], got " + s);
        auto parse_usize = [](const std::string &t, uint64_t *v) {      // Rust str::parse::<usize>
            size_t i = (!t.empty() && t[0] == '+') ? 1 : 0;
            if (i >= t.size()) return false;
            *v = 0;
            for (; i < t.size(); i++) {
                if (t[i] < '0' || t[i] > '9') return false;
                const uint64_t dgt = (uint64_t)(t[i] - '0');
                if (*v > (0xffffffffffffffffULL - dgt) / 10) return false;      // overflow of a 64-bit usize: Err like Rust
                *v = *v * 10 + dgt;
            }
            return true;
        };
        uint64_t kn = 0, kf = 0, km = 16;
        double rf = 0.0;
        if (!parse_usize(parts[0], &kn)) return fail(SR_ERR_INVALID, "Invalid neighbor count: " + parts[0]);
        if (parts.size() >= 2 && !parse_usize(parts[1], &kf)) return fail(SR_ERR_INVALID, "Invalid stranger count: " + parts[1]);
        if (parts.size() >= 3) {
            if (!parse_f64(parts[2], &rf)) return fail(SR_ERR_INVALID, "Invalid random fraction: " + parts[2]);
            if (rf < 0.0 || rf > 1.0) return fail(SR_ERR_INVALID, "Random fraction must be in [0.0, 1.0], got " + parts[2]);
        }
        if (parts.size() >= 4) {
            if (!parse_usize(parts[3], &km)) return fail(SR_ERR_INVALID, "Invalid k-mer size: " + parts[3]);
            if (km == 0) return fail(SR_ERR_INVALID, "K-mer size must be > 0");
        }
        p->sparsify_kind = SR_SPARSE_TREE;
        p->tree_k_nearest = (uint32_t)std::min<uint64_t>(kn, 0xffffffffULL);
        p->tree_k_farthest = (uint32_t)std::min<uint64_t>(kf, 0xffffffffULL);
        p->tree_rand_frac = rf; p->tree_kmer = (uint32_t)std::min<uint64_t>(km, 0xffffffffULL);
        return SR_OK;
    }
    if (parse_f64(s, &f) && f > 0.0 && f <= 1.0) { p->sparsify_kind = SR_SPARSE_RANDOM; p->sparsify_factor = f; return SR_OK; }
    return fail(SR_ERR_INVALID, "Invalid sparsification: '" + s + "'");
}

// max_score_for_divergence seqrush.rs:253-269
static int32_t max_score_for_divergence(const sr_params &p, uint64_t seq_len, double d) {
    int32_t max_mismatches = (int32_t)std::ceil((double)seq_len * d);
    int32_t max_gaps = (int32_t)std::ceil((double)seq_len * d * 0.5);
    int32_t mismatch_score = max_mismatches * p.mismatch_penalty;
    int32_t gap_score = max_gaps > 0 ? p.gap_open1 + (max_gaps - 1) * p.gap_ext1 : 0;
    return std::max(mismatch_score + gap_score, p.mismatch_penalty * 2);
}

// ------------------------------------------------------------------ context
struct sr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    sr_params prm;
    uint32_t n = 0;
    std::vector<uint32_t> len;
    std::vector<uint64_t> goff;
    uint64_t total_len = 0, uf_size = 0;
    std::vector<uint32_t> pair_q, pair_t;          // this rank's pairs, in enumeration order
    std::vector<uint64_t> cigar_base;              // [np + 1] prefix sums of the per-pair CIGAR reserve (ops)
    std::vector<uint32_t> batch_first;             // [nbatch + 1]: pair ranges that share the CIGAR arena one after the other
    uint64_t total_pairs_all_ranks = 0;            // size of the (sparsified) list before sharding
    uint64_t dp_cells = 0;
    int off16 = 1, nwg = 0, nthreads = 256, symbits = 2;
    size_t lds_bytes = 0;
    SrAlignArgs aa{};
    SrUniteArgs ua{};
    std::vector<void *> dev_allocs;
    unsigned long long *d_nodes = nullptr, *d_minarr = nullptr, *d_labels = nullptr;
    unsigned long long *d_counters = nullptr;
    int *d_error = nullptr;
    uint32_t *d_queue = nullptr, *d_oqueue = nullptr, *d_order = nullptr;
    uint64_t *d_okeys = nullptr, *d_okeys2 = nullptr;   // cost-ordered dequeue after the orientation kernel (sr_order.hip)
    uint32_t *d_ovals = nullptr; void *d_otemp = nullptr; size_t otemp_bytes = 0;
    uint64_t *d_cbase = nullptr;                   // [np + nbatch]: batch b starts at batch_first[b] + b, values relative to its arena
    uint8_t *d_bases = nullptr;                    // raw bytes of all sequences (sketching, graph induction)
    int onwg = 0;                      // workgroups (= waves) of the orientation kernel, 0 = orientation inside the alignment kernel
    size_t olds_bytes = 0;
    int32_t *d_max_score = nullptr;
    // kernel timing: [kind] -> one event pair per batch (kinds: 0 align, 1 unite, 4 orientation) or a single pair (2, 3)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[5];
    int ev_used[5] = {0, 0, 0, 0, 0};
    bool loaded = false;
    bool fuse_unite = false;           // sr_ctx_run: the blocked alignment kernel unites each pair after its CIGAR (no unite kernel)
    bool from_paf = false;             // loaded by sr_ctx_load_paf: no alignment stage, no sr_alignments
    bool aligned_batch_valid = false;  // the arena holds the CIGARs of the last batch sr_ctx_align ran
    std::string workspace_report;      // sizing of the last load (sr_ctx_workspace_report)
};

static int dev_alloc(sr_ctx *c, void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        char buf[160];
        snprintf(buf, sizeof(buf), "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return fail(e == hipErrorOutOfMemory ? SR_ERR_NOMEM : SR_ERR_HIP, buf);
    }
    c->dev_allocs.push_back(*p);
    return SR_OK;
}
static void free_dev(sr_ctx *c) {
    for (void *p : c->dev_allocs) (void)hipFree(p);
    c->dev_allocs.clear();
    c->d_nodes = c->d_minarr = c->d_labels = c->d_counters = nullptr;
    c->d_error = nullptr; c->d_queue = nullptr; c->d_oqueue = nullptr; c->d_order = nullptr; c->d_cbase = nullptr;
    c->d_bases = nullptr; c->d_max_score = nullptr; c->onwg = 0;
    c->d_okeys = c->d_okeys2 = nullptr; c->d_ovals = nullptr; c->d_otemp = nullptr; c->otemp_bytes = 0;
    c->loaded = false; c->from_paf = false; c->aligned_batch_valid = false;
    for (int k = 0; k < 5; k++) c->ev_used[k] = 0;
}

// event pair `idx` of kind `kind` (created on first use)
static int ev_get(sr_ctx *c, int kind, int idx, hipEvent_t **a, hipEvent_t **b) {
    while ((int)c->ev[kind].size() <= idx) {
        hipEvent_t x = nullptr, y = nullptr;
        HIPCHK(hipEventCreate(&x)); HIPCHK(hipEventCreate(&y));
        c->ev[kind].push_back({x, y});
    }
    *a = &c->ev[kind][idx].first; *b = &c->ev[kind][idx].second;
    return SR_OK;
}

extern "C" int sr_ctx_create(int device, sr_ctx **out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(SR_ERR_NO_DEVICE, "no HIP device available (seqrush_amd has no CPU fallback)");
    if (device < 0 || device >= n) return fail(SR_ERR_INVALID, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    sr_ctx *c = new sr_ctx();
    c->device = device;
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return fail(SR_ERR_HIP, "hipStreamCreate failed"); }
    c->own_stream = true;
    *out = c;
    return SR_OK;
}

extern "C" void sr_ctx_destroy(sr_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_dev(c);
    for (int k = 0; k < 5; k++)
        for (auto &e : c->ev[k]) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int sr_ctx_set_stream(sr_ctx *c, void *hip_stream) {
    if (!c) return fail(SR_ERR_INVALID, "null ctx");
    if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return SR_OK;
}

static inline uint8_t comp_base(uint8_t b) {      // Sequence::reverse_complement seqrush.rs:281-295
    switch (b) {
    case 'A': case 'a': return 'T'; case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C';
    default: return b;
    }
}

static uint64_t splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
static inline double unit53(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

// ------------------------------------------------------------------ pair list (A2)
// Ordered pair list before sharding: row-major n*n incl. self pairs ("all-vs-all including self",
// seqrush.rs:718-734), optional exclude_self, then the sparsification strategy (grammar seqrush.rs:356-431).
// allwave's selection rules are not in the reference tree; the definitions below are this project's own
// (restated in oracle/seqrush.c, PARITY UNPINNED):
//   random:F        ordered pair (q,t), q != t, kept iff unit(splitmix64(seed ^ (q*n+t))) < F
//   connectivity:P  unordered pair {i<j} kept iff unit(splitmix64(seed ^ (i*n+j))) < f,
//                   f = min(1, (ln n + c) / n), c = -ln(-ln P): the Erdos-Renyi threshold at which G(n,f) is
//                   connected with probability P (P = 1 or n <= 2: f = 1); both directions of a kept pair are aligned
//   auto            n < 10: none; else connectivity:0.99
//   tree:kn,kf,rf,k unordered pair kept iff it is one of the kn nearest or kf farthest neighbours of either end by
//                   k-mer sketch similarity (sr_sketch.hip), or unit(splitmix64(seed ^ (i*n+j))) < rf
// Self pairs are always kept unless exclude_self.
static double connectivity_fraction(uint32_t n, double P) {
    if (n <= 2 || P >= 1.0) return 1.0;
    const double c = -std::log(-std::log(P));
    const double f = (std::log((double)n) + c) / (double)n;
    return f >= 1.0 ? 1.0 : (f <= 0.0 ? 0.0 : f);
}

// sel: n*n bytes from the device k-NN selection (tree), or NULL
static void enumerate_pairs(uint32_t n, const sr_params &p, const uint8_t *sel, std::vector<uint32_t> &pq, std::vector<uint32_t> &pt) {
    pq.clear(); pt.clear();
    int kind = p.sparsify_kind;
    double frac = p.sparsify_factor;
    if (kind == SR_SPARSE_AUTO) { if (n < 10) kind = SR_SPARSE_NONE; else { kind = SR_SPARSE_CONNECTIVITY; frac = 0.99; } }
    const double conn_f = (kind == SR_SPARSE_CONNECTIVITY) ? connectivity_fraction(n, frac) : 1.0;
    for (uint32_t q = 0; q < n; q++)
        for (uint32_t t = 0; t < n; t++) {
            if (q == t) { if (p.exclude_self) continue; pq.push_back(q); pt.push_back(t); continue; }
            const uint32_t i = std::min(q, t), j = std::max(q, t);
            bool keep = true;
            if (kind == SR_SPARSE_RANDOM)
                keep = unit53(splitmix64(p.sparsify_seed ^ ((uint64_t)q * n + t))) < frac;
            else if (kind == SR_SPARSE_CONNECTIVITY)
                keep = unit53(splitmix64(p.sparsify_seed ^ ((uint64_t)i * n + j))) < conn_f;
            else if (kind == SR_SPARSE_TREE)
                keep = (sel && (sel[(uint64_t)i * n + j] || sel[(uint64_t)j * n + i])) ||
                       unit53(splitmix64(p.sparsify_seed ^ ((uint64_t)i * n + j))) < p.tree_rand_frac;
            if (keep) { pq.push_back(q); pt.push_back(t); }
        }
}

// Cost-balanced shard: pairs sorted by cost (|q|*|t|; self pairs cost |q|: they end in one
// extension), longest first, each handed to the least loaded rank (lowest rank on ties); a rank keeps its pairs
// in enumeration order.  Every rank computes the same assignment.  Equal costs degenerate to round-robin.
static void shard_pairs(const std::vector<uint32_t> &len, uint32_t rank, uint32_t count, std::vector<uint32_t> &pq,
                        std::vector<uint32_t> &pt) {
    if (count <= 1) return;
    const size_t m = pq.size();
    std::vector<uint64_t> cost(m);
    for (size_t i = 0; i < m; i++)
        cost[i] = pq[i] == pt[i] ? (uint64_t)len[pq[i]] : (uint64_t)len[pq[i]] * (uint64_t)len[pt[i]];
    std::vector<uint32_t> idx(m);
    for (size_t i = 0; i < m; i++) idx[i] = (uint32_t)i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    typedef std::pair<uint64_t, uint32_t> LR;           // (load, rank): min-heap
    std::priority_queue<LR, std::vector<LR>, std::greater<LR>> heap;
    for (uint32_t r = 0; r < count; r++) heap.push({0, r});
    std::vector<uint8_t> mine(m, 0);
    for (uint32_t i : idx) {
        LR top = heap.top(); heap.pop();
        if (top.second == rank) mine[i] = 1;
        top.first += cost[i] ? cost[i] : 1;
        heap.push(top);
    }
    size_t w = 0;
    for (size_t i = 0; i < m; i++) if (mine[i]) { pq[w] = pq[i]; pt[w] = pt[i]; w++; }
    pq.resize(w); pt.resize(w);
}

extern "C" int sr_pair_list(uint32_t n, const sr_params *p, uint32_t **q_out, uint32_t **t_out, uint64_t *count) {
    if (!p || !q_out || !t_out || !count) return fail(SR_ERR_INVALID, "null argument");
    if (p->shard_count == 0 || p->shard_rank >= p->shard_count) return fail(SR_ERR_INVALID, "bad shard");
    if (p->sparsify_kind == SR_SPARSE_TREE)
        return fail(SR_ERR_UNSUPPORTED, "tree sparsification needs the sequences (k-mer sketches): load a context and call sr_ctx_pairs");
    if ((uint64_t)n * n > 0xffffffffULL) return fail(SR_ERR_UNSUPPORTED, "more than 2^32 ordered pairs");
    std::vector<uint32_t> pq, pt;
    enumerate_pairs(n, *p, nullptr, pq, pt);
    // sharding by cost needs lengths; this host-only helper assumes equal lengths (= round-robin over the
    // off-diagonal pairs, then over the cheaper self pairs)
    std::vector<uint32_t> len(n, 2);
    shard_pairs(len, p->shard_rank, p->shard_count, pq, pt);
    *count = pq.size();
    *q_out = (uint32_t *)malloc((pq.size() ? pq.size() : 1) * 4);
    *t_out = (uint32_t *)malloc((pt.size() ? pt.size() : 1) * 4);
    memcpy(*q_out, pq.data(), pq.size() * 4);
    memcpy(*t_out, pt.data(), pt.size() * 4);
    return SR_OK;
}

static int make_pen(const sr_params &p, bool ori, SrPen *out) {
    if (ori) {
        if (p.ori_match != 0) return fail(SR_ERR_UNSUPPORTED, "orientation match score must be 0");
        out->x = p.ori_mismatch; out->o1 = p.ori_gap_open; out->e1 = p.ori_gap_ext;
        out->o2 = 0; out->e2 = 0; out->two = 0;
    } else {
        if (p.match_score != 0)
            return fail(SR_ERR_UNSUPPORTED, "match score must be 0 (WFA2 penalties, seqrush.rs:45); the Aligner-trait defaults "
                                            "2,4,4,2,24,1 (src/aligner/allwave_impl.rs:15-23) need allwave's score conversion, "
                                            "which is not in the reference tree -- see INTEGRATION.md section 8");
        out->x = p.mismatch_penalty; out->o1 = p.gap_open1; out->e1 = p.gap_ext1;
        out->two = p.gap_open2 >= 0;
        out->o2 = out->two ? p.gap_open2 : 0; out->e2 = out->two ? p.gap_ext2 : 0;
        if (out->two && out->e2 <= 0) return fail(SR_ERR_INVALID, "gap2_extend must be > 0");
    }
    if (out->x <= 0 || out->o1 < 0 || out->e1 <= 0) return fail(SR_ERR_INVALID, "penalties must satisfy x>0, o>=0, e>0");
    out->scope = std::max(out->x, out->o1 + out->e1);
    if (out->two) out->scope = std::max(out->scope, out->o2 + out->e2);
    out->scope += 1;
    if (out->scope > SR_MAX_SCOPE) return fail(SR_ERR_UNSUPPORTED, "penalties too large for the device ring (scope > 127)");
    return SR_OK;
}

// ------------------------------------------------------------------ load
// Symbol codes of the packed buffer.  The reference compares raw bytes (seqrush.rs:1162-1176, 1268-1283), so any
// injective byte -> code map keeps every comparison: 2 bits when the inputs (and their complements) are within
// upper-case ACGT, 4 bits for <= 16 distinct bytes (N, IUPAC, soft-masked lower case), else the bytes themselves.
struct SymMap { int bits; int code[256]; };
static SymMap make_symmap(const sr_seqset *seqs) {
    bool present[256] = {false};
    const uint64_t N = seqs->offsets[seqs->n];
    for (uint64_t i = 0; i < N; i++) present[seqs->bases[i]] = true;
    for (int b = 0; b < 256; b++) if (present[b]) present[comp_base((uint8_t)b)] = true;
    SymMap m;
    for (int b = 0; b < 256; b++) m.code[b] = -1;
    bool acgt = true;
    int distinct = 0;
    for (int b = 0; b < 256; b++) if (present[b]) { distinct++; if (b != 'A' && b != 'C' && b != 'G' && b != 'T') acgt = false; }
    if (acgt) { m.bits = 2; m.code['A'] = 0; m.code['C'] = 1; m.code['G'] = 2; m.code['T'] = 3; return m; }
    if (distinct <= 16) { m.bits = 4; int k = 0; for (int b = 0; b < 256; b++) if (present[b]) m.code[b] = k++; return m; }
    m.bits = 8;
    for (int b = 0; b < 256; b++) m.code[b] = b;
    return m;
}

// A load runs these steps in order (round 3: one function each):
//   load_index        lengths, offsets, penalties, parameter checks
//   pack_sequences    byte -> symbol map, packed copies (forward / reverse complement [/ reversed / complemented])
//   upload_sequences  raw bytes + packed words + per-sequence tables to the device
//   build_pair_list   enumeration or explicit list, sparsification (tree: k-mer sketches on the device), shard
//   plan_launch       which kernel, workgroup shape, per-workgroup workspace, CIGAR arena and batches, workgroup count
//   alloc_workspace   device allocations, per-pair arrays, kernel arguments, sizing report
struct PackedSeqs {
    SymMap sm;
    int per_word = 16, ncopies = 2;
    std::vector<uint64_t> woff[4];                 // word offset of every sequence's copy k (k = 2, 3 alias 1, 0 for 2-bit buffers)
    std::vector<uint32_t> words;
};
struct SeqDev { uint32_t *words = nullptr; uint64_t *woff[4] = {nullptr, nullptr, nullptr, nullptr}; uint64_t *goff = nullptr; uint32_t *len = nullptr; };
struct Plan {
    // kernel + launch shape
    int impl = 1, kblock = 0, wg_per_cu = 4, cus = 256, lazy_id = 0, ring_u16 = 0, bbase_jobs = 16;
    bool wave_wg = false;
    uint32_t max_words = 0;
    size_t osz = 2, rsz = 2;
    // per-workgroup workspace (cells unless stated)
    int ring_scope = 0, ring_hot = 0, hist_levels = 0, hist_w = 0, brow = 0, kdepth = 0, kdepth2 = 0;
    uint64_t bring_wg = 0, bhist_wg = 0, hist_cap = 0, hist_nul_w = 0, bseg_wg = 0, bbt_wg = 0, bcl_wg = 0, per_wg_bytes = 0;
    // memory budget
    size_t free_b = 0;
    uint64_t arena_ops = 0, left = 0, max_reserve = 4;
    uint32_t nbatch = 0, max_batch_pairs = 0;
    int nwg = 1;
    uint64_t oring_bytes = 0;
};
#define DEV_UPLOAD(field, T, hostvec)                                                         \
    do {                                                                                      \
        const auto &hv_ = (hostvec);                                                          \
        void *d_ = nullptr;                                                                   \
        if ((r = dev_alloc(c, &d_, hv_.size() * sizeof(T)))) return r;                        \
        HIPCHK(hipMemcpyAsync(d_, hv_.data(), hv_.size() * sizeof(T), hipMemcpyHostToDevice, c->stream)); \
        field = (T *)d_;                                                                      \
    } while (0)

static int load_index(sr_ctx *c, const sr_seqset *seqs, const sr_params *p, bool explicit_pairs, SrPen *pen, SrPen *ori, uint64_t *maxlen_out) {
    c->prm = *p;
    const uint32_t n = seqs->n;
    c->n = n;
    c->len.assign(n, 0); c->goff.assign(n, 0);
    uint64_t maxlen = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t L = seqs->offsets[i + 1] - seqs->offsets[i];
        if (L == 0) {
            std::string nm = (seqs->names && seqs->names[i]) ? seqs->names[i] : std::to_string(i);
            return fail(SR_ERR_EMPTY_SEQ, "Empty sequences are not allowed: sequence '" + nm + "' has length 0");
        }
        if (L > 0x7fff0000ULL) return fail(SR_ERR_UNSUPPORTED, "sequence too long");
        c->len[i] = (uint32_t)L; c->goff[i] = seqs->offsets[i];
        maxlen = std::max(maxlen, L);
    }
    *maxlen_out = maxlen;
    c->total_len = seqs->offsets[n];
    c->uf_size = (c->total_len << 1) + 2;          // bidirected_union_find.rs:16-24
    int r;
    if ((r = make_pen(*p, false, pen))) return r;
    if ((r = make_pen(*p, true, ori))) return r;
    if (p->memory_mode != SR_MEM_ULTRALOW)
        return fail(SR_ERR_UNSUPPORTED, "memory_mode must be SR_MEM_ULTRALOW (biWFA, what the reference selects: src/wfa.rs:57); "
                                        "the device keeps full wavefront history only for biWFA's base cases");
    if (p->sparsify_kind < SR_SPARSE_NONE || p->sparsify_kind > SR_SPARSE_TREE) return fail(SR_ERR_INVALID, "bad sparsify_kind");
    if (p->sparsify_kind == SR_SPARSE_TREE && (p->tree_kmer < 1 || p->tree_kmer > 32))
        return fail(SR_ERR_UNSUPPORTED, "tree sparsification: k-mer size must be in 1..32 on the device");
    if (p->shard_count == 0 || p->shard_rank >= p->shard_count) return fail(SR_ERR_INVALID, "bad shard");
    if (!explicit_pairs && (uint64_t)n * n > 0xffffffffULL)
        return fail(SR_ERR_UNSUPPORTED, "more than 2^32 ordered pairs: sparsify (-x) or pass an explicit pair list");
    return SR_OK;
}

// packed symbol buffer: forward, reverse complement (+ reversed, complemented when the alphabet is not plain ACGT: seqrush's
// complement maps lower case to upper case, so "equal complements <=> equal bases" fails); one pad word either side of a copy
static void pack_sequences(const sr_ctx *c, const sr_seqset *seqs, PackedSeqs &pk) {
    const uint32_t n = seqs->n;
    pk.sm = make_symmap(seqs);
    pk.per_word = 32 / pk.sm.bits;
    pk.ncopies = pk.sm.bits == 2 ? 2 : 4;
    for (int k = 0; k < 4; k++) pk.woff[k].assign(n, 0);
    uint64_t words = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t w = (c->len[i] + pk.per_word - 1) / pk.per_word;
        for (int k = 0; k < pk.ncopies; k++) { pk.woff[k][i] = words + 1; words += w + 2; }
        if (pk.ncopies == 2) { pk.woff[2][i] = pk.woff[1][i]; pk.woff[3][i] = pk.woff[0][i]; }      // aliases (see sr_align_blk.inc)
    }
    pk.words.assign(words, 0);
    const SymMap &sm = pk.sm;
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t *b = seqs->bases + seqs->offsets[i];
        const uint64_t L = c->len[i];
        for (uint64_t j = 0; j < L; j++) {
            const unsigned sh = (unsigned)(j % pk.per_word) * sm.bits;
            const uint64_t wi = j / pk.per_word;
            pk.words[pk.woff[0][i] + wi] |= (uint32_t)sm.code[b[j]] << sh;                          // forward
            pk.words[pk.woff[1][i] + wi] |= (uint32_t)sm.code[comp_base(b[L - 1 - j])] << sh;      // reverse complement
            if (pk.ncopies == 4) {
                pk.words[pk.woff[2][i] + wi] |= (uint32_t)sm.code[b[L - 1 - j]] << sh;              // reversed
                pk.words[pk.woff[3][i] + wi] |= (uint32_t)sm.code[comp_base(b[j])] << sh;           // complemented
            }
        }
    }
}

static int upload_sequences(sr_ctx *c, const sr_seqset *seqs, const PackedSeqs &pk, SeqDev &sd) {
    int r;
    void *d;
    if ((r = dev_alloc(c, &d, c->total_len))) return r;                // raw bytes (sketches, graph induction)
    c->d_bases = (uint8_t *)d;
    HIPCHK(hipMemcpyAsync(c->d_bases, seqs->bases, c->total_len, hipMemcpyHostToDevice, c->stream));
    DEV_UPLOAD(sd.words, uint32_t, pk.words);
    DEV_UPLOAD(sd.woff[0], uint64_t, pk.woff[0]);
    DEV_UPLOAD(sd.woff[1], uint64_t, pk.woff[1]);
    if (pk.ncopies == 4) { DEV_UPLOAD(sd.woff[2], uint64_t, pk.woff[2]); DEV_UPLOAD(sd.woff[3], uint64_t, pk.woff[3]); }
    else { sd.woff[2] = sd.woff[1]; sd.woff[3] = sd.woff[0]; }
    DEV_UPLOAD(sd.len, uint32_t, c->len);
    DEV_UPLOAD(sd.goff, uint64_t, c->goff);
    HIPCHK(hipStreamSynchronize(c->stream));                           // (the host vectors may go away after this)
    return SR_OK;
}

// `tree:` sparsification: bottom-1000 sketches, pairwise similarities and the k-nearest / k-farthest selection on the device
static int tree_selection(sr_ctx *c, const sr_params *p, const SeqDev &sd, std::vector<uint8_t> &sel) {
    const uint32_t n = c->n;
    const int S = 1000;
    std::vector<uint32_t> npad(n);
    uint64_t stride = 2;
    for (uint32_t i = 0; i < n; i++) { uint32_t v = 2; while (v < c->len[i]) v <<= 1; npad[i] = v; stride = std::max<uint64_t>(stride, v); }
    struct Tmp { std::vector<void *> v; ~Tmp() { for (void *q : v) (void)hipFree(q); } } tmp;
    auto talloc = [&](size_t bytes) -> void * { void *q = nullptr; if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr; tmp.v.push_back(q); return q; };
    unsigned long long *d_scr = (unsigned long long *)talloc((size_t)n * stride * 8);
    unsigned long long *d_sk = (unsigned long long *)talloc((size_t)n * S * 8);
    uint32_t *d_skn = (uint32_t *)talloc((size_t)n * 4), *d_npad = (uint32_t *)talloc((size_t)n * 4);
    uint32_t *d_sh = (uint32_t *)talloc((size_t)n * n * 4), *d_dn = (uint32_t *)talloc((size_t)n * n * 4);
    uint8_t *d_sel = (uint8_t *)talloc((size_t)n * n);
    if (!d_scr || !d_sk || !d_skn || !d_npad || !d_sh || !d_dn || !d_sel) return fail(SR_ERR_NOMEM, "not enough device memory for the k-mer sketches");
    HIPCHK(hipMemcpyAsync(d_npad, npad.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(d_sel, 0, (size_t)n * n, c->stream));
    HIPCHK(hipMemsetAsync(d_sh, 0, (size_t)n * n * 4, c->stream));
    HIPCHK(hipMemsetAsync(d_dn, 0, (size_t)n * n * 4, c->stream));
    if (srk_sketch(c->d_bases, sd.goff, sd.len, n, (int)p->tree_kmer, S, d_scr, stride, d_npad, d_sk, d_skn, c->stream) ||
        srk_jaccard(d_sk, d_skn, n, S, d_sh, d_dn, c->stream) ||
        srk_knn_select(d_sh, d_dn, n, (int)std::min<uint32_t>(p->tree_k_nearest, n), (int)std::min<uint32_t>(p->tree_k_farthest, n), d_sel, c->stream))
        return fail(SR_ERR_HIP, "sketch kernel launch failed");
    sel.resize((size_t)n * n);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(sel.data(), d_sel, (size_t)n * n, hipMemcpyDeviceToHost));
    return SR_OK;
}

// ordered pair list of this rank + what follows from it: DP cells, worst-case CIGAR reserve per pair, divergence thresholds
static int build_pair_list(sr_ctx *c, const sr_params *p, const SeqDev &sd, const uint32_t *eq, const uint32_t *et, uint64_t ecount,
                           bool explicit_pairs, std::vector<int32_t> &max_score, uint64_t *max_reserve) {
    const uint32_t n = c->n;
    if (explicit_pairs) {
        c->pair_q.assign(eq, eq + ecount); c->pair_t.assign(et, et + ecount);
        for (uint64_t i = 0; i < ecount; i++)
            if (eq[i] >= n || et[i] >= n) return fail(SR_ERR_INVALID, "pair index out of range");
    } else {
        std::vector<uint8_t> sel;
        if (p->sparsify_kind == SR_SPARSE_TREE && n > 1) { const int r = tree_selection(c, p, sd, sel); if (r) return r; }
        enumerate_pairs(n, *p, sel.empty() ? nullptr : sel.data(), c->pair_q, c->pair_t);
    }
    c->total_pairs_all_ranks = c->pair_q.size();
    shard_pairs(c->len, p->shard_rank, p->shard_count, c->pair_q, c->pair_t);
    if (c->pair_q.size() > 0xfffffff0ULL) return fail(SR_ERR_UNSUPPORTED, "more than 2^32 pairs in one shard");
    const uint32_t np = (uint32_t)c->pair_q.size();
    c->dp_cells = 0;
    c->cigar_base.assign((size_t)np + 1, 0);
    max_score.assign(std::max<uint32_t>(np, 1), INT_MAX);
    *max_reserve = 4;
    for (uint32_t i = 0; i < np; i++) {
        const uint64_t lq = c->len[c->pair_q[i]], lt = c->len[c->pair_t[i]];
        c->dp_cells += lq * lt;
        c->cigar_base[i + 1] = c->cigar_base[i] + lq + lt + 2;
        *max_reserve = std::max(*max_reserve, lq + lt + 2);
        if (p->max_divergence >= 0.0) max_score[i] = max_score_for_divergence(*p, std::min(lq, lt), p->max_divergence);
    }
    return SR_OK;
}

// kernel choice and workgroup shape.  impl 2 = score-blocked wave-tiled kernel (when this build has an instance for the
// penalties), impl 1 = level-synchronous kernel (any penalties; rings deeper than 32 levels take its wide instance)
static int plan_kernel(sr_ctx *c, const PackedSeqs &pk, const SrPen &pen, const SrPen &ori, uint64_t maxlen, uint32_t np, Plan &pl) {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    pl.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int cus = pl.cus, bits = pk.sm.bits;
    c->off16 = maxlen <= 32000 ? 1 : 0;
    if (knob("SR_FORCE_INT32")) c->off16 = 0;             // (tests: the 32-bit searches and their 16-bit ring on short sequences)
    pl.osz = c->off16 ? 2 : 4;
    pl.max_words = (uint32_t)((maxlen + pk.per_word - 1) / pk.per_word + 2);
    c->lds_bytes = (size_t)pl.max_words * 3 * 4;          // query fwd / rc + target fwd
    if ((long long)c->lds_bytes > (long long)srk_align_max_lds())
        return fail(SR_ERR_UNSUPPORTED, "sequences too long to stage in LDS (limit ~ 170 kb per sequence at 2 bits per base)");
    pl.impl = 1;
    pl.kblock = srk_align_blk_supports(&pen, &ori);
    // SR_BLK_LEVELS=5: the generic 5-level instance instead of the exact-penalty 10-level one (A/B runs, tests)
    if (const char *e = knob("SR_BLK_LEVELS")) { if (atoi(e) == 5 && pl.kblock == 10) pl.kblock = 5; }
    // (the blocked kernel stages four sequence copies, has ~20 KB of static LDS and keeps its LDS window addresses within
    // 128 KB: static tables + copies must fit below; it has its own ring depth limit, srk_align_blk_supports)
    if (pl.kblock > 0 && (long long)pl.max_words * 16 + 28 * 1024 <= 128 * 1024) pl.impl = 2;
    if (const char *e = knob("SR_ALIGN_IMPL")) { if (atoi(e) <= 1) pl.impl = 1; }      // (tests: the level-per-pass kernel)
    int &impl = pl.impl, &kblock = pl.kblock;
    if (impl == 2) c->lds_bytes = (size_t)pl.max_words * 4 * 4 + 16;      // (+ read slack of a two-window extension step)
    pl.wg_per_cu = 4;
    if (const char *e = knob("SR_WG_PER_CU")) pl.wg_per_cu = std::max(1, atoi(e));
    c->nthreads = 256;
    // few pairs (e.g. the 1/8 shard of C2): one pair per workgroup leaves CUs short of waves, so give every
    // pair 8 waves instead of 4 (measured 31 -> 23 ms for 529 pairs of 5 kb)
    if (impl == 2 && (uint64_t)np <= 2ULL * (uint64_t)cus + (uint64_t)cus / 2) c->nthreads = 512;
    // fewer pairs than CUs (C3: 144): a pair has a CU to itself -- 16 waves (C3 90.1 -> 69.2 ms)
    if (impl == 2 && (uint64_t)np <= (uint64_t)cus) c->nthreads = 1024;
    if (const char *e = knob("SR_ALIGN_THREADS")) {
        const int v = atoi(e);
        if (v == 128 || v == 256 || v == 512 || (v == 1024 && impl == 2) || (v == 64 && impl == 2 && bits == 2) ||
            (v == 192 && impl == 2 && kblock == 10 && c->off16 && bits == 2 && pen.two)) c->nthreads = v;   // (192: experiment builds only, -DSR_NT192)
    }
    if (c->nthreads == 1024 && !(impl == 2 && kblock == 10 && c->off16 && bits == 2 && pen.two)) c->nthreads = 512;   // (the one 1024-thread build)
    if (impl == 2 && c->nthreads == 128 && kblock == 10 && !(bits == 2 && c->off16)) kblock = 5;   // (no 128-thread 10-level instance there)
    pl.wave_wg = impl == 2 && (c->nthreads == 64 || (c->nthreads == 128 && kblock == 10));   // lean builds: 16 / 8 pairs per CU
    if (pl.wave_wg) pl.wg_per_cu = c->nthreads == 64 ? 16 : 8;
    // static tables of the kernel (upper estimates; the blocked kernel's scan arrays of the fused unite grow with the
    // workgroup: 23 416 / 26 536 / 32 776 bytes at 256 / 512 / 1024 threads)
    auto static_lds = [&](int nthreads) -> size_t {
        if (const char *e = knob("SR_STATIC_LDS_KB")) return (size_t)std::max(1, atoi(e)) * 1024;       // (A/B builds with other table sizes)
        return (size_t)(pl.wave_wg ? 6 : impl != 2 ? 28 : nthreads >= 1024 ? 33 : nthreads >= 512 ? 26 : 23) * 1024;
    };
    const size_t lds_per_wg = c->lds_bytes + static_lds(c->nthreads);
    pl.wg_per_cu = (int)std::min<size_t>((size_t)pl.wg_per_cu, std::max<size_t>(1, (160 * 1024) / lds_per_wg));
    // long sequences: the four LDS copies leave room for two workgroups per CU (C5: 50 KB each) -- give each 8 waves, or
    // the SIMDs hold two waves (C5 subset 3 507 -> 2 200 ms).  512-thread builds: int16 rows; 32-bit searches with the
    // uint16 ring (exact instance below 57 k, see ring_u16)
    const char *ru = knob("SR_RING_U16");
    const bool u16_ok = kblock == 10 && maxlen <= 57000 && !(ru && atoi(ru) == 0);
    if (impl == 2 && !pl.wave_wg && pl.wg_per_cu <= 2 && !knob("SR_ALIGN_THREADS") && (c->off16 || u16_ok)) c->nthreads = 512;
    pl.wg_per_cu = (int)std::min<size_t>((size_t)pl.wg_per_cu, std::max<size_t>(1, (160 * 1024) / (c->lds_bytes + static_lds(c->nthreads))));
    // 32-bit searches below 57 k keep their ring as uint16 (offset + 8192): half the row bytes (C5 is bound by them).
    // The exact 10-level instance at >= 256 threads has that build; SR_RING_U16=0 keeps 32-bit rows.
    pl.ring_u16 = (impl == 2 && !c->off16 && u16_ok && c->nthreads >= 256) ? 1 : 0;
    pl.rsz = pl.ring_u16 ? 2 : pl.osz;                 // bytes per ring cell (the base-case history keeps osz)
    pl.bbase_jobs = pl.wave_wg ? (c->nthreads == 64 ? 4 : 8) : 16;
    if (const char *e = knob("SR_BFS_BASE_JOBS")) pl.bbase_jobs = std::max(1, std::min(16, atoi(e)));
    return SR_OK;
}

// per-workgroup workspace: shared rows (every aligner of a pass owns a sub-range), base-case history, segment lists,
// reversed op buffers of the base cases, breakpoint candidate list, per-level maxima
static int plan_workspace(const sr_ctx *c, const SrPen &pen, const SrPen &ori, uint64_t maxlen, Plan &pl) {
    const int impl = pl.impl, kblock = pl.kblock;
    pl.ring_scope = std::max(pen.scope, ori.scope);
    int emax = std::max(pen.e1, ori.e1);
    if (pen.two) emax = std::max(emax, pen.e2);
    pl.ring_hot = emax + 2;
    const int gapmax = pen.two ? std::max(pen.o1, pen.o2) : pen.o1;
    auto gc = [&](int len) { int g = pen.o1 + pen.e1 * len; if (pen.two) g = std::min(g, pen.o2 + pen.e2 * len); return g; };
    // worst base case: score <= 250 (+ one gap-open: a half that ends in a gap component) or both lengths <= 100
    const int smax_base = std::max(250 + gapmax, 2 * gc(100)) + 2 * gapmax + 4;
    pl.hist_levels = smax_base + 1 + std::max(kblock, 5);   // + one block of levels (impl 2 computes whole blocks)
    int rmax = smax_base / pen.e1;
    if (pen.two) rmax = std::max(rmax, smax_base / pen.e2);
    pl.hist_w = (2 * (rmax + pen.scope + 2) + 1 + 8 + 64 + 7) & ~7;   // rows hold whole 4-diagonal groups, + halo groups either side
    pl.brow = (int)((4 * maxlen + 32 * 72 + 64 + 512 + 7) & ~7ULL);     // per-aligner margins + read slack of the last wave tile
    pl.kdepth = std::max(pen.scope + std::max(kblock, 1), ori.scope + 1) + 1;
    // lazy I/D rows (sr_align_blk.inc blk_recompute): the M rows must reach 2 * scope + 2 blocks back
    const char *lz = knob("SR_LAZY_ID");
    pl.lazy_id = (impl == 2 && !(lz && atoi(lz) == 0) && 2 * pen.scope + 2 * kblock + 2 <= SR_BLK_MAK_SLOTS) ? 1 : 0;
    pl.kdepth2 = pl.kdepth;
    if (pl.lazy_id) {
        // I / D rows are read by breakpoint detection (scope levels back from the newest block) and written by the recompute
        // pass from the block boundary before that window: scope + 2 blocks + 2 levels are live at most
        pl.kdepth2 = std::max(pl.kdepth, pen.scope + 2 * kblock + 2);
        pl.kdepth = std::max(pl.kdepth, 2 * pen.scope + 2 * kblock + 2);
    }
    if (impl == 2 && kblock == 10) {
        // exact instance: ring depths are multiples of the block, so that a block's ten levels (and five levels from a multiple
        // of five) are adjacent rows that never straddle the ring's wrap -- the tile addresses them as base + immediate
        // (sr_align_blk.inc kbase; default penalties: 74 -> 80 M levels, 48 -> 50 I / D levels)
        pl.kdepth = (pl.kdepth + 9) / 10 * 10; pl.kdepth2 = (pl.kdepth2 + 9) / 10 * 10;
    }
    // level-per-pass kernel: M ring | 4 hot I/D rings | cold I/D history | NULL row | U row, rows of brow cells
    pl.bring_wg = ((uint64_t)(pl.ring_scope + 1) + 4ULL * pl.ring_hot + 4ULL * (pl.ring_scope + 1) + 2ULL) * (uint64_t)pl.brow;
    // blocked kernel: chunk-major ring (sr_align_blk.inc KRows): 256-cell pieces of all depth + 4 * depth2 + 3 rows together
    // (NULL, U, trash), + 2 pieces of read slack; a workgroup's rows are addressed as base + 32-bit byte offset
    if (impl == 2) pl.bring_wg = ((uint64_t)pl.brow / 256 + 2ULL) * ((uint64_t)pl.kdepth + 4ULL * pl.kdepth2 + (uint64_t)SR_NULL_ROWS + 2ULL) * 256ULL + 1024;
    if (impl == 2 && pl.bring_wg * pl.rsz >= (1ULL << 32)) return fail(SR_ERR_UNSUPPORTED, "sequences too long for the device row workspace (4 GB per workgroup)");
    // base-case history.  Level-per-pass kernel: bbase_jobs fixed slots of the worst-case width.  Blocked kernel: hist_cap
    // cells in which every job of a batch gets a region for the levels and width it really needs (sr_align_blk.inc); one
    // worst-case job always fits, the default holds 4 of them (C2: 11 MB instead of 40.6 -- a pair's 16 base cases of score
    // ~150 need 6.5 M cells and still run as one batch; SR_HIST_JOBS=n: n of them)
    const uint64_t hist_worst = (uint64_t)pl.hist_levels * 5 * (uint64_t)pl.hist_w;
    pl.hist_cap = hist_worst * 4;
    if (const char *e = knob("SR_HIST_JOBS")) pl.hist_cap = hist_worst * (uint64_t)std::max(1, std::min(16, atoi(e)));
    pl.hist_nul_w = impl == 2 ? (uint64_t)pl.hist_w + 256 : (uint64_t)pl.bbase_jobs * (uint64_t)pl.hist_w;
    pl.bhist_wg = impl == 2 ? ((pl.hist_cap + pl.hist_nul_w + 256 + 1024 + 7) & ~7ULL)      // data, NULL row, trash cells, slack
                            : ((uint64_t)pl.hist_levels * 5 + 1) * pl.hist_nul_w + 1024;
    pl.bseg_wg = 2ULL * SR_BFS_MAXSEG * SR_BFS_SEGREC * 4;          // bytes
    pl.bbt_wg = (uint64_t)pl.bbase_jobs * SR_BFS_BTCAP * 4;        // bytes
    pl.bcl_wg = 0;                 // (round 3: breakpoint detection keeps a band per unit in LDS instead of a candidate list in global memory)
    pl.per_wg_bytes = pl.bring_wg * pl.rsz + pl.bhist_wg * pl.osz + pl.bseg_wg + pl.bbt_wg + pl.bcl_wg * 4 + (impl == 2 ? 32 * SR_BLK_MAK_SLOTS * 4 : 0);
    (void)c;
    return SR_OK;
}

// memory budget: every buffer counted.  fixed = union-find arrays + per-pair arrays; then the CIGAR arena (worst
// case |q|+|t|+2 ops per pair; pairs run in batches that reuse it), the orientation rings and the alignment workspaces share
// what is left
static int plan_memory(sr_ctx *c, uint32_t np, Plan &pl) {
    size_t total_b = 0;
    HIPCHK(hipMemGetInfo(&pl.free_b, &total_b));
    const uint64_t fixed = 3ULL * c->uf_size * 8 + (uint64_t)np * 40 + (1ULL << 20);
    if (fixed + (64ULL << 20) > pl.free_b) return fail(SR_ERR_NOMEM, "not enough device memory for the union-find arrays");
    const uint64_t avail = pl.free_b - fixed;
    pl.arena_ops = c->cigar_base[np] + 1;
    {
        uint64_t cap = std::max<uint64_t>((uint64_t)(avail * 0.25) / 4, pl.max_reserve + 1);
        if (const char *e = knob("SR_CIGAR_ARENA_OPS")) cap = std::max<uint64_t>((uint64_t)atoll(e), pl.max_reserve + 1);   // (tests: force batches)
        pl.arena_ops = std::min(pl.arena_ops, cap);
    }
    if (pl.arena_ops * 4 > avail / 2) return fail(SR_ERR_NOMEM, "not enough device memory for one pair's CIGAR");
    c->batch_first.assign(1, 0);
    for (uint32_t i = 0; i < np; i++)
        if (c->cigar_base[i + 1] - c->cigar_base[c->batch_first.back()] + 1 > pl.arena_ops) c->batch_first.push_back(i);
    c->batch_first.push_back(np);
    pl.nbatch = (uint32_t)c->batch_first.size() - 1;
    pl.max_batch_pairs = 0;
    for (uint32_t b = 0; b < pl.nbatch; b++) pl.max_batch_pairs = std::max(pl.max_batch_pairs, c->batch_first[b + 1] - c->batch_first[b]);
    pl.left = avail - pl.arena_ops * 4;
    const uint64_t budget = (uint64_t)(pl.left * 0.7);
    int nwg = pl.cus * pl.wg_per_cu;
    if (const char *e = knob("SR_NWG")) nwg = std::max(1, std::min(nwg, atoi(e)));   // (tests: several pairs per workgroup on a small input)
    if ((uint64_t)nwg > pl.max_batch_pairs) nwg = (int)pl.max_batch_pairs;
    if (nwg < 1) nwg = 1;
    while (nwg > 1 && (uint64_t)nwg * pl.per_wg_bytes > budget) nwg--;
    if ((uint64_t)nwg * pl.per_wg_bytes > budget) return fail(SR_ERR_NOMEM, "not enough device memory for one workgroup's wavefront ring");
    pl.nwg = c->nwg = nwg;
    return SR_OK;
}

// per-pair device arrays, workspaces, kernel arguments
static int alloc_workspace(sr_ctx *c, const sr_params *p, const PackedSeqs &pk, const SeqDev &sd, const SrPen &pen, const SrPen &ori,
                           uint64_t maxlen, const std::vector<int32_t> &max_score, Plan &pl) {
    int r;
    void *d;
    const uint32_t np = (uint32_t)c->pair_q.size(), nbatch = pl.nbatch;
    const int impl = pl.impl, nwg = pl.nwg;
    const size_t osz = pl.osz, rsz = pl.rsz;
    SrAlignArgs &a = c->aa;
    memset(&a, 0, sizeof(a));
    uint32_t *d_pq, *d_pt;
    std::vector<uint32_t> pq = c->pair_q, pt = c->pair_t;
    if (pq.empty()) { pq.push_back(0); pt.push_back(0); }
    DEV_UPLOAD(d_pq, uint32_t, pq);
    DEV_UPLOAD(d_pt, uint32_t, pt);
    // per-batch relative CIGAR bases: batch b owns entries [first_b + b, first_b + b + count_b]
    std::vector<uint64_t> cb((size_t)np + nbatch, 0);
    for (uint32_t b = 0; b < nbatch; b++) {
        const uint32_t f = c->batch_first[b], l = c->batch_first[b + 1];
        for (uint32_t i = f; i <= l; i++) cb[(size_t)i + b] = c->cigar_base[i] - c->cigar_base[f];
    }
    DEV_UPLOAD(c->d_cbase, uint64_t, cb);
    // cost-sorted dequeue order inside each batch: the longest pairs start first (self pairs last)
    std::vector<uint32_t> order(std::max<uint32_t>(np, 1), 0);
    for (uint32_t b = 0; b < nbatch; b++) {
        const uint32_t f = c->batch_first[b], l = c->batch_first[b + 1];
        std::vector<uint32_t> idx(l - f);
        for (uint32_t i = 0; i < l - f; i++) idx[i] = i;
        auto cost = [&](uint32_t i) { const uint32_t q = c->pair_q[f + i], t = c->pair_t[f + i];
                                      return q == t ? (uint64_t)c->len[q] : (uint64_t)c->len[q] * c->len[t]; };
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return cost(x) > cost(y); });
        for (uint32_t i = 0; i < l - f; i++) order[f + i] = idx[i];
    }
    DEV_UPLOAD(c->d_order, uint32_t, order);
    DEV_UPLOAD(c->d_max_score, int32_t, max_score);
    HIPCHK(hipStreamSynchronize(c->stream));   // the uploads above have read their host vectors (pq, pt, cb, order)
    if ((r = dev_alloc(c, &d, sizeof(uint32_t)))) return r; c->d_queue = (uint32_t *)d;
    if ((r = dev_alloc(c, &d, (uint64_t)nwg * pl.bring_wg * rsz))) return r; a.bring = d;
    if ((r = dev_alloc(c, &d, (uint64_t)nwg * pl.bhist_wg * osz))) return r; a.bhist = d;
    // (tests: the kernels must not depend on what the row workspaces held before -- poison them with plausible offsets)
    if (const char *e = knob("SR_POISON_ROWS")) {
        const int v = atoi(e);
        if (rsz == 2) HIPCHK(hipMemsetD16Async((hipDeviceptr_t)a.bring, (unsigned short)(pl.ring_u16 ? v - 24576 : v), (size_t)nwg * pl.bring_wg, c->stream));
        else HIPCHK(hipMemsetD32Async((hipDeviceptr_t)a.bring, v, (size_t)nwg * pl.bring_wg, c->stream));
        if (osz == 2) HIPCHK(hipMemsetD16Async((hipDeviceptr_t)a.bhist, (unsigned short)v, (size_t)nwg * pl.bhist_wg, c->stream));
        else HIPCHK(hipMemsetD32Async((hipDeviceptr_t)a.bhist, v, (size_t)nwg * pl.bhist_wg, c->stream));
    }
    if ((r = dev_alloc(c, &d, (uint64_t)nwg * pl.bseg_wg))) return r; a.bseg = (int *)d;
    if ((r = dev_alloc(c, &d, (uint64_t)nwg * pl.bbt_wg))) return r; a.bbt = (uint32_t *)d;
    if (pl.bcl_wg) { if ((r = dev_alloc(c, &d, (uint64_t)nwg * pl.bcl_wg * 4))) return r; a.bcl = (uint32_t *)d; a.bcl_wg_stride = pl.bcl_wg; }
    if (impl == 2) { if ((r = dev_alloc(c, &d, (uint64_t)nwg * 32 * SR_BLK_MAK_SLOTS * 4))) return r; a.bmak = (int *)d; }
    // orientation as its own kernel, one pair per wave (sr_orient.hip); SR_PREORIENT=0 keeps it in the alignment kernel.
    // Worth it when there are enough pairs to fill the chip with one wave each and the three sequence copies of a
    // wave leave >= 8 waves per CU (measured: C2 / C4 faster; C3 -- 144 pairs -- 1.6x and C5 -- 50 kb, 4 waves per
    // CU by LDS -- 9 % slower than orientation inside the alignment kernel's workgroup).  SR_PREORIENT=1 / 0 forces.
    const char *po = knob("SR_PREORIENT");
    const bool pre_auto = (uint64_t)np >= 4ULL * (uint64_t)pl.cus && (size_t)pl.max_words * 12 <= 20 * 1024;
    if (impl == 2 && (po ? atoi(po) != 0 : pre_auto) && (size_t)pl.max_words * 12 <= 60 * 1024) {
        const int orow = (int)((2 * ((2 * maxlen + 32) & ~3ULL) + 512 + 7) & ~7ULL);
        const uint64_t oring_wg = ((uint64_t)(ori.scope + 1) * 3 + 1) * (uint64_t)orow + 256;
        int onwg = (int)std::min<uint64_t>(pl.max_batch_pairs, (uint64_t)pl.cus * 16);
        while (onwg > 1 && (uint64_t)onwg * oring_wg * osz > (uint64_t)(pl.left * 0.2)) onwg--;
        if (onwg >= 1 && np > 0) {
            pl.oring_bytes = (uint64_t)onwg * oring_wg * osz;
            if ((r = dev_alloc(c, &d, pl.oring_bytes))) return r; a.oring = d;
            if ((r = dev_alloc(c, &d, sizeof(uint32_t)))) return r; c->d_oqueue = (uint32_t *)d;
            a.oring_wg_stride = oring_wg; a.orow = orow; a.oqueue = c->d_oqueue; a.pre_oriented = 1;
            if (!knob("SR_NO_REORDER")) {       // dequeue order from the orientation scores, per batch
                c->otemp_bytes = srk_order_temp_bytes(pl.max_batch_pairs);
                if ((r = dev_alloc(c, &d, (uint64_t)pl.max_batch_pairs * 8))) return r; c->d_okeys = (uint64_t *)d;
                if ((r = dev_alloc(c, &d, (uint64_t)pl.max_batch_pairs * 8))) return r; c->d_okeys2 = (uint64_t *)d;
                if ((r = dev_alloc(c, &d, (uint64_t)pl.max_batch_pairs * 4))) return r; c->d_ovals = (uint32_t *)d;
                if ((r = dev_alloc(c, &d, std::max<size_t>(c->otemp_bytes, 16)))) return r; c->d_otemp = d;
            }
            c->onwg = onwg; c->olds_bytes = (size_t)pl.max_words * 3 * 4;
        }
    }
    if ((r = dev_alloc(c, &d, (size_t)np + 1))) return r; a.is_reverse = (uint8_t *)d;
    if ((r = dev_alloc(c, &d, ((size_t)np + 1) * 4))) return r; a.score = (int32_t *)d;
    if ((r = dev_alloc(c, &d, ((size_t)np + 1) * 4))) return r; a.ori_fwd = (int32_t *)d;
    if ((r = dev_alloc(c, &d, ((size_t)np + 1) * 4))) return r; a.ori_rev = (int32_t *)d;
    if ((r = dev_alloc(c, &d, ((size_t)np + 1) * 4))) return r; a.cigar_cnt = (uint32_t *)d;
    if ((r = dev_alloc(c, &d, (pl.arena_ops + 1) * 4))) return r; a.cigar_ops = (uint32_t *)d;
    if ((r = dev_alloc(c, &d, SR_NCOUNTERS * sizeof(unsigned long long)))) return r; c->d_counters = (unsigned long long *)d;
    if ((r = dev_alloc(c, &d, sizeof(int)))) return r; c->d_error = (int *)d;
    if ((r = dev_alloc(c, &d, c->uf_size * 8))) return r; c->d_nodes = (unsigned long long *)d;
    if ((r = dev_alloc(c, &d, c->uf_size * 8))) return r; c->d_minarr = (unsigned long long *)d;
    if ((r = dev_alloc(c, &d, c->uf_size * 8))) return r; c->d_labels = (unsigned long long *)d;
    HIPCHK(hipMemsetAsync(c->d_counters, 0, SR_NCOUNTERS * sizeof(unsigned long long), c->stream));
    HIPCHK(hipMemsetAsync(c->d_error, 0, sizeof(int), c->stream));
    HIPCHK(hipMemsetAsync(a.cigar_cnt, 0, ((size_t)np + 1) * 4, c->stream));
    HIPCHK(hipMemsetAsync(a.score, 0xff, ((size_t)np + 1) * 4, c->stream));
    a.seqwords = sd.words; a.word_off_fwd = sd.woff[0]; a.word_off_rc = sd.woff[1]; a.word_off_rev = sd.woff[2]; a.word_off_cmp = sd.woff[3];
    a.symbits = pk.sm.bits; a.order = c->d_order; a.seqlen = sd.len;
    a.max_words = pl.max_words; a.pair_q = d_pq; a.pair_t = d_pt; a.npairs = np;
    a.queue_head = c->d_queue; a.pen = pen; a.ori = ori; a.mem_mode = p->memory_mode;
    a.ring_scope = pl.ring_scope; a.ring_hot = pl.ring_hot;
    a.hist_w = pl.hist_w; a.hist_levels = pl.hist_levels;
    { const char *pt_ = knob("SR_PROFILE_TICKS"); a.profile_ticks = (pt_ && atoi(pt_) != 0) ? 1 : 0; }
    a.impl = impl; a.kdepth = pl.kdepth; a.kdepth2 = pl.kdepth2; a.kblock = pl.kblock; a.lazy_id = pl.lazy_id; a.ring_u16 = pl.ring_u16;
    a.ori_levels = knob("SR_ORIENT_LEVELS") ? 1 : 0;
    { const char *tb = knob("SR_TEST_BASE_LEVELS"); a.test_base_levels = tb ? std::max(0, atoi(tb)) : 0; }
    a.bring_wg_stride = pl.bring_wg; a.brow = pl.brow; a.bhist_wg_stride = pl.bhist_wg; a.bbase_jobs = pl.bbase_jobs;
    a.hist_cap = pl.hist_cap; a.hist_nul_w = (uint32_t)pl.hist_nul_w; a.hist_stride = (uint32_t)((uint64_t)pl.bbase_jobs * (uint64_t)pl.hist_w);
    a.cigar_base = c->d_cbase; a.counters = c->d_counters; a.error_flag = c->d_error;
    // 8-mer sets of the sequences (8 KB each) for the orientation kernel's lower bound of the reverse orientation's score:
    // plain ACGT buffers, default orientation penalties (the blocked orientation kernel), at most 256 MB and 5 % of what
    // is left.  SR_NO_KBITS=1 switches the bound off (both orientations in lockstep from level 0, as before round 3).
    if (a.pre_oriented && pk.sm.bits == 2 && !ori.two && ori.x == 1 && ori.o1 == 1 && ori.e1 == 1 && !knob("SR_NO_KBITS") && !knob("SR_ORIENT_LEVELS")) {
        const uint64_t kb = (uint64_t)c->len.size() * 8192ull;
        if (kb > 0 && kb <= (256ull << 20) && kb <= (uint64_t)(pl.left * 0.05)) {
            if ((r = dev_alloc(c, &d, kb))) return r;
            if ((r = srk_kmer_bits(&a, (uint32_t)c->len.size(), (uint32_t *)d, c->stream))) return fail(SR_ERR_DEVICE_FAULT, "k-mer set kernel failed to launch");
            a.kbits = (const uint32_t *)d;
        }
    }
    SrUniteArgs &u = c->ua;
    memset(&u, 0, sizeof(u));
    u.pair_q = d_pq; u.pair_t = d_pt; u.npairs = np; u.seqlen = sd.len; u.seq_goff = sd.goff;
    a.fuse_unite = 0; a.uf_nodes = c->d_nodes; a.seq_goff = sd.goff; a.max_score = c->d_max_score; a.min_match_len = p->min_match_len;
    c->fuse_unite = impl == 2 && !knob("SR_NO_FUSED_UNITE");
    u.is_reverse = a.is_reverse; u.score = a.score; u.max_score = c->d_max_score;
    u.cigar_ops = a.cigar_ops; u.cigar_base = c->d_cbase; u.cigar_cnt = a.cigar_cnt;
    u.min_match_len = p->min_match_len; u.nodes = c->d_nodes; u.uf_size = c->uf_size;
    u.counters = c->d_counters; u.error_flag = c->d_error;
    return SR_OK;
}

static void write_report(sr_ctx *c, const PackedSeqs &pk, const SrPen &pen, const Plan &pl) {
    const uint32_t np = (uint32_t)c->pair_q.size();
    char buf[1500];
    snprintf(buf, sizeof(buf),
             "{\"pairs\": %u, \"batches\": %u, \"symbol_bits\": %d, \"offset_bytes\": %zu, \"ring_cell_bytes\": %zu, \"kernel_impl\": %d, "
             "\"block_levels\": %d, \"two_piece\": %d, \"lazy_id_rows\": %d, \"workgroups\": %d, "
             "\"threads_per_workgroup\": %d, \"workgroups_per_cu\": %d, \"lds_dynamic_bytes\": %zu, \"ring_bytes_per_workgroup\": %llu, "
             "\"base_history_bytes_per_workgroup\": %llu, \"workspace_bytes\": %llu, \"cigar_arena_bytes\": %llu, "
             "\"orientation_ring_bytes\": %llu, \"union_find_bytes\": %llu, \"device_free_bytes_at_load\": %zu, \"kernel_build\": \"%s\", "
             "\"ring_depth_m\": %d, \"ring_depth_id\": %d, \"fused_unite\": %d, \"source_digest\": \"%s\", \"knobs\": ",
             np, pl.nbatch, pk.sm.bits, pl.osz, pl.impl == 2 ? pl.rsz : pl.osz, pl.impl, pl.impl == 2 ? pl.kblock : 1, pen.two ? 1 : 0, pl.lazy_id,
             pl.nwg, c->nthreads, pl.wg_per_cu, c->lds_bytes,
             (unsigned long long)(pl.bring_wg * pl.rsz), (unsigned long long)(pl.bhist_wg * pl.osz),
             (unsigned long long)((uint64_t)pl.nwg * pl.per_wg_bytes), (unsigned long long)(pl.arena_ops * 4), (unsigned long long)pl.oring_bytes,
             (unsigned long long)(3ULL * c->uf_size * 8), pl.free_b, srk_align_blk_build_tag(), pl.kdepth, pl.kdepth2, c->fuse_unite ? 1 : 0, srk_source_digest());
    c->workspace_report = std::string(buf) + knobs_json() + "}";
}

static int load_impl(sr_ctx *c, const sr_seqset *seqs, const sr_params *p, const uint32_t *eq, const uint32_t *et,
                     uint64_t ecount, bool explicit_pairs) {
    if (!c || !seqs || !p) return fail(SR_ERR_INVALID, "null argument");
    if (seqs->n == 0) return fail(SR_ERR_INVALID, "no sequences");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    free_dev(c);
    int r;
    SrPen pen, ori;
    uint64_t maxlen = 0;
    if ((r = load_index(c, seqs, p, explicit_pairs, &pen, &ori, &maxlen))) return r;
    PackedSeqs pk;
    pack_sequences(c, seqs, pk);
    c->symbits = pk.sm.bits;
    SeqDev sd;
    if ((r = upload_sequences(c, seqs, pk, sd))) return r;
    Plan pl;
    std::vector<int32_t> max_score;
    if ((r = build_pair_list(c, p, sd, eq, et, ecount, explicit_pairs, max_score, &pl.max_reserve))) return r;
    const uint32_t np = (uint32_t)c->pair_q.size();
    if ((r = plan_kernel(c, pk, pen, ori, maxlen, np, pl))) return r;
    if ((r = plan_workspace(c, pen, ori, maxlen, pl))) return r;
    if ((r = plan_memory(c, np, pl))) return r;
    if ((r = alloc_workspace(c, p, pk, sd, pen, ori, maxlen, max_score, pl))) return r;
    write_report(c, pk, pen, pl);
    if (srk_uf_init(c->d_nodes, c->total_len, c->uf_size, c->stream)) return fail(SR_ERR_HIP, "uf init launch failed");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->loaded = true;
    return SR_OK;
}
#undef DEV_UPLOAD

extern "C" int sr_ctx_load(sr_ctx *c, const sr_seqset *seqs, const sr_params *p) {
    return load_impl(c, seqs, p, nullptr, nullptr, 0, false);
}
extern "C" int sr_ctx_load_pairs(sr_ctx *c, const sr_seqset *seqs, const sr_params *p, const uint32_t *query_idx,
                                 const uint32_t *target_idx, uint64_t count) {
    if (count && (!query_idx || !target_idx)) return fail(SR_ERR_INVALID, "null pair list");
    return load_impl(c, seqs, p, query_idx, target_idx, count, true);
}

extern "C" int sr_ctx_pairs(const sr_ctx *c, uint32_t **q_out, uint32_t **t_out, uint64_t *count) {
    if (!c || !c->loaded || !q_out || !t_out || !count) return fail(SR_ERR_INVALID, "context not loaded");
    const size_t m = c->pair_q.size();
    *count = m;
    *q_out = (uint32_t *)malloc((m ? m : 1) * 4);
    *t_out = (uint32_t *)malloc((m ? m : 1) * 4);
    memcpy(*q_out, c->pair_q.data(), m * 4);
    memcpy(*t_out, c->pair_t.data(), m * 4);
    return SR_OK;
}
extern "C" uint32_t sr_ctx_num_batches(const sr_ctx *c) { return (c && c->loaded) ? (uint32_t)c->batch_first.size() - 1 : 0; }
extern "C" const char *sr_ctx_workspace_report(const sr_ctx *c) { return c ? c->workspace_report.c_str() : ""; }

extern "C" int sr_ctx_reset_uf(sr_ctx *c) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    HIPCHK(hipSetDevice(c->device));
    if (srk_uf_init(c->d_nodes, c->total_len, c->uf_size, c->stream)) return fail(SR_ERR_HIP, "uf init launch failed");
    return SR_OK;
}

// kernel arguments of batch b: the per-pair arrays shifted to the batch, its slice of d_cbase
static void batch_args(const sr_ctx *c, uint32_t b, SrAlignArgs *a, SrUniteArgs *u) {
    const uint32_t f = c->batch_first[b], cnt = c->batch_first[b + 1] - f;
    if (a) {
        *a = c->aa;
        a->pair_q += f; a->pair_t += f; a->npairs = cnt; if (a->order) a->order += f;
        a->is_reverse += f; a->score += f; a->ori_fwd += f; a->ori_rev += f; a->cigar_cnt += f;
        if (a->max_score) a->max_score += f;
        a->cigar_base = c->d_cbase + f + b;
    }
    if (u) {
        *u = c->ua;
        u->pair_q += f; u->pair_t += f; u->npairs = cnt; u->is_reverse += f; u->score += f; u->max_score += f;
        u->cigar_cnt += f; u->cigar_base = c->d_cbase + f + b;
        if (u->q_start) { u->q_start += f; u->t_start += f; }
    }
}

static int enqueue_align_batch(sr_ctx *c, uint32_t b, bool fuse_unite = false) {
    SrAlignArgs a;
    batch_args(c, b, &a, nullptr);
    a.fuse_unite = fuse_unite ? 1 : 0;
    HIPCHK(hipMemsetAsync(c->d_queue, 0, sizeof(uint32_t), c->stream));
    hipEvent_t *e0, *e1;
    int r;
    if (c->onwg > 0 && a.npairs > 0) {           // which = 4: the orientation kernel
        HIPCHK(hipMemsetAsync(c->d_oqueue, 0, sizeof(uint32_t), c->stream));
        if ((r = ev_get(c, 4, c->ev_used[4], &e0, &e1))) return r;
        HIPCHK(hipEventRecord(*e0, c->stream));
        r = srk_orient(&a, std::min<int>(c->onwg, (int)a.npairs), c->olds_bytes, c->off16, c->stream);
        if (r) return fail(SR_ERR_HIP, std::string("orientation kernel launch failed: ") + hipGetErrorString((hipError_t)r));
        if (c->d_okeys && a.order) {             // longest predicted alignment first (orientation score x length)
            r = srk_order(&a, c->d_okeys, c->d_okeys2, c->d_ovals, c->d_otemp, c->otemp_bytes, (uint32_t *)a.order, c->stream);
            if (r) return fail(SR_ERR_HIP, std::string("dequeue-order sort failed: ") + hipGetErrorString((hipError_t)r));
        }
        HIPCHK(hipEventRecord(*e1, c->stream));
        c->ev_used[4]++;
    }
    if ((r = ev_get(c, 0, c->ev_used[0], &e0, &e1))) return r;
    HIPCHK(hipEventRecord(*e0, c->stream));
    if (a.npairs > 0) {
        r = srk_align(&a, std::min<int>(c->nwg, (int)a.npairs), c->lds_bytes, c->off16, c->nthreads, c->stream);
        if (r) return fail(SR_ERR_HIP, std::string("align kernel launch failed: ") + hipGetErrorString((hipError_t)r));
    }
    HIPCHK(hipEventRecord(*e1, c->stream));
    c->ev_used[0]++;
    return SR_OK;
}

static int enqueue_unite_batch(sr_ctx *c, uint32_t b, bool fused = false) {
    SrUniteArgs u;
    batch_args(c, b, nullptr, &u);
    hipEvent_t *e0, *e1;
    int r;
    if ((r = ev_get(c, 1, c->ev_used[1], &e0, &e1))) return r;
    HIPCHK(hipEventRecord(*e0, c->stream));
    if (u.npairs > 0 && !fused) {                  // (fused: the alignment kernel did it; the empty event pair keeps kernel_ms(1) defined)
        int nwg = (int)std::min<uint64_t>(u.npairs, 4096);
        r = srk_unite(&u, nwg, c->stream);
        if (r) return fail(SR_ERR_HIP, std::string("unite kernel launch failed: ") + hipGetErrorString((hipError_t)r));
    }
    HIPCHK(hipEventRecord(*e1, c->stream));
    c->ev_used[1]++;
    return SR_OK;
}

static int begin_pass(sr_ctx *c, bool reset_counters) {
    HIPCHK(hipSetDevice(c->device));
    if (reset_counters) {
        HIPCHK(hipMemsetAsync(c->d_counters, 0, SR_NCOUNTERS * sizeof(unsigned long long), c->stream));
        c->ev_used[0] = c->ev_used[4] = 0;
    }
    return SR_OK;
}

extern "C" int sr_ctx_align(sr_ctx *c) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    if (c->from_paf) return fail(SR_ERR_INVALID, "context was loaded from a PAF file: there is no alignment stage");
    if (c->batch_first.size() > 2)
        return fail(SR_ERR_UNSUPPORTED, "the pair list runs in several batches that share one CIGAR arena: use sr_ctx_run "
                                        "(align + unite per batch) or sr_align_all");
    int r = begin_pass(c, true);
    if (r) return r;
    if ((r = enqueue_align_batch(c, 0))) return r;
    c->aligned_batch_valid = true;
    return SR_OK;
}

extern "C" int sr_ctx_unite(sr_ctx *c) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    if (c->batch_first.size() > 2) return fail(SR_ERR_UNSUPPORTED, "several batches: use sr_ctx_run");
    HIPCHK(hipSetDevice(c->device));
    c->ev_used[1] = 0;
    return enqueue_unite_batch(c, 0);
}

// alignment + match-run extraction + unite of the whole pair list, batch after batch (no host sync)
extern "C" int sr_ctx_run(sr_ctx *c) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    int r = begin_pass(c, !c->from_paf);
    if (r) return r;
    c->ev_used[1] = 0;
    const uint32_t nbatch = (uint32_t)c->batch_first.size() - 1;
    const bool fuse = c->fuse_unite && !c->from_paf;
    for (uint32_t b = 0; b < nbatch; b++) {
        if (!c->from_paf && (r = enqueue_align_batch(c, b, fuse))) return r;
        if ((r = enqueue_unite_batch(c, b, fuse))) return r;
    }
    c->aligned_batch_valid = nbatch == 1 && !c->from_paf;
    return SR_OK;
}

extern "C" int sr_ctx_sync(sr_ctx *c) {
    if (!c) return fail(SR_ERR_INVALID, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->loaded) {
        int err = 0;
        HIPCHK(hipMemcpy(&err, c->d_error, sizeof(int), hipMemcpyDeviceToHost));
        if (err) {
            static const char *names[9] = {"score bound exceeded", "base-case history overflow", "backtrace found no predecessor",
                                           "recursion stack / segment list overflow", "CIGAR buffer overflow", "union-find retry bound",
                                           "breakpoint outside its segment", "graph induction",
                                           "row / LDS address outside the workgroup's extent (bounds-checked build)"};
            std::string msg = "device reported internal fault bits";
            char buf[32];
            snprintf(buf, sizeof(buf), " 0x%x:", err);
            msg += buf;
            for (int b = 0; b < 9; b++) if (err & (1 << b)) { msg += " ["; msg += names[b]; msg += "]"; }
            if (err & SR_DEV_ERR_ADDRESS) {                  // first offender: counters[40..43] = pair + 1, what, offset, extent
                uint64_t d[4] = {0, 0, 0, 0};
                if (hipMemcpy(d, c->d_counters + 40, sizeof(d), hipMemcpyDeviceToHost) == hipSuccess && d[0]) {
                    char b2[160];
                    snprintf(b2, sizeof(b2), " first: pair #%llu, %s, offset %llu, extent %llu", (unsigned long long)(d[0] - 1),
                             d[1] == 1 ? "ring row" : d[1] == 2 ? "history row" : "LDS window", (unsigned long long)d[2], (unsigned long long)d[3]);
                    msg += b2;
                }
            }
            // which pair(s): a failed alignment leaves score -1
            if (!c->from_paf && c->aa.score) {
                const size_t np = c->pair_q.size();
                std::vector<int32_t> sc(np);
                if (np && hipMemcpy(sc.data(), c->aa.score, np * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                    int shown = 0;
                    for (size_t i = 0; i < np && shown < 4; i++)
                        if (sc[i] < 0) { snprintf(buf, sizeof(buf), " pair(%u,%u)", c->pair_q[i], c->pair_t[i]); msg += buf; shown++; }
                }
            }
            return fail(SR_ERR_DEVICE_FAULT, msg);
        }
    }
    return SR_OK;
}

extern "C" uint64_t sr_ctx_uf_size(const sr_ctx *c) { return c ? c->uf_size : 0; }
extern "C" uint64_t sr_ctx_num_pairs(const sr_ctx *c) { return c ? c->pair_q.size() : 0; }
extern "C" uint64_t sr_ctx_dp_cells(const sr_ctx *c) { return c ? c->dp_cells : 0; }

extern "C" const char *sr_ctx_align_kernel(const sr_ctx *c) {
    if (!c || !c->loaded || c->from_paf) return nullptr;
    return c->aa.impl == 2 ? "sr_align_blk_kernel" : "sr_align_bfs_kernel";
}

extern "C" int sr_ctx_kernel_ms(sr_ctx *c, int which, float *ms) {
    if (!c || which < 0 || which > 4 || c->ev_used[which] == 0) return fail(SR_ERR_INVALID, "no timing recorded");
    float tot = 0;
    for (int i = 0; i < c->ev_used[which]; i++) {
        float t = 0;
        HIPCHK(hipEventSynchronize(c->ev[which][i].second));
        HIPCHK(hipEventElapsedTime(&t, c->ev[which][i].first, c->ev[which][i].second));
        tot += t;
    }
    *ms = tot;
    return SR_OK;
}

extern "C" int sr_ctx_counters(sr_ctx *c, uint64_t out[16]) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->d_counters, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SR_OK;
}

extern "C" int sr_ctx_counters_ext(sr_ctx *c, uint64_t out[32]) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->d_counters, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SR_OK;
}
// every slot the device keeps (SR_NCOUNTERS = 48 since round 4); returns the number of slots written (<= cap) or a negative status
extern "C" int sr_ctx_counters_all(sr_ctx *c, uint64_t *out, uint32_t cap) {
    if (!c || !c->loaded || !out) return fail(SR_ERR_INVALID, "context not loaded");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t n = std::min<uint32_t>(cap, SR_NCOUNTERS);
    HIPCHK(hipMemcpy(out, c->d_counters, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return (int)n;
}

extern "C" int sr_ctx_download_uf(sr_ctx *c, uint64_t *parent_out) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(parent_out, c->d_nodes, c->uf_size * 8, hipMemcpyDeviceToHost));
    return SR_OK;
}

static int labels_timed(sr_ctx *c, uint64_t *dev_labels64, uint32_t *dev_labels32) {
    hipEvent_t *e0, *e1;
    int r;
    if ((r = ev_get(c, 2, 0, &e0, &e1))) return r;
    HIPCHK(hipEventRecord(*e0, c->stream));
    if (dev_labels64) {
        if (srk_labels(c->d_nodes, c->uf_size, c->d_minarr, (unsigned long long *)dev_labels64, c->d_error, c->stream))
            return fail(SR_ERR_HIP, "label kernel launch failed");
    } else {
        if (srk_labels32(c->d_nodes, c->uf_size, c->d_minarr, dev_labels32, c->d_error, c->stream))
            return fail(SR_ERR_HIP, "label kernel launch failed");
    }
    HIPCHK(hipEventRecord(*e1, c->stream));
    c->ev_used[2] = 1;
    return SR_OK;
}

extern "C" int sr_ctx_labels_device(sr_ctx *c, uint64_t *dev_labels) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    HIPCHK(hipSetDevice(c->device));
    return labels_timed(c, dev_labels, nullptr);
}
extern "C" int sr_ctx_labels_device_u32(sr_ctx *c, uint32_t *dev_labels) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    if (c->uf_size > 0xffffffffULL) return fail(SR_ERR_UNSUPPORTED, "2N+2 >= 2^32: use the 64-bit label exchange");
    HIPCHK(hipSetDevice(c->device));
    return labels_timed(c, nullptr, dev_labels);
}

extern "C" int sr_ctx_merge_labels(sr_ctx *c, const uint64_t *dev_labels, uint32_t count) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    HIPCHK(hipSetDevice(c->device));
    if (srk_merge(c->d_nodes, c->uf_size, (const unsigned long long *)dev_labels, count, c->d_error, c->stream))
        return fail(SR_ERR_HIP, "merge kernel launch failed");
    return SR_OK;
}
extern "C" int sr_ctx_merge_labels_u32(sr_ctx *c, const uint32_t *dev_labels, uint32_t count) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    if (c->uf_size > 0xffffffffULL) return fail(SR_ERR_UNSUPPORTED, "2N+2 >= 2^32: use the 64-bit label exchange");
    HIPCHK(hipSetDevice(c->device));
    if (srk_merge32(c->d_nodes, c->uf_size, dev_labels, count, c->d_error, c->stream))
        return fail(SR_ERR_HIP, "merge kernel launch failed");
    return SR_OK;
}

// host-resident label arrays (e.g. read back from files written by other processes / nodes): upload + replay-unite
extern "C" int sr_ctx_merge_labels_host(sr_ctx *c, const uint64_t *labels, uint32_t count) {
    if (!c || !c->loaded || !labels) return fail(SR_ERR_INVALID, "context not loaded");
    HIPCHK(hipSetDevice(c->device));
    for (uint32_t i = 0; i < count; i++) {
        HIPCHK(hipMemcpyAsync(c->d_labels, labels + (uint64_t)i * c->uf_size, c->uf_size * 8, hipMemcpyHostToDevice, c->stream));
        if (srk_merge(c->d_nodes, c->uf_size, c->d_labels, 1, c->d_error, c->stream)) return fail(SR_ERR_HIP, "merge kernel launch failed");
    }
    return SR_OK;
}

extern "C" int sr_ctx_download_labels(sr_ctx *c, uint64_t *labels_out) {
    int r = sr_ctx_labels_device(c, (uint64_t *)c->d_labels);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(labels_out, c->d_labels, c->uf_size * 8, hipMemcpyDeviceToHost));
    return SR_OK;
}

// ------------------------------------------------------------------ results
extern "C" void sr_alignments_free(sr_alignments *a) {
    if (!a) return;
    free(a->query_idx); free(a->target_idx); free(a->is_reverse); free(a->score);
    free(a->query_start); free(a->query_end); free(a->target_start); free(a->target_end);
    free(a->cigar_off); free(a->cigar_ops);
    free(a);
}

// host copy of the alignments of batch b (its CIGARs must still be in the arena), appended to `ops`
struct AlnAcc { std::vector<uint32_t> cnt, ops; std::vector<int32_t> score; std::vector<uint8_t> isrev; };
static int collect_batch(sr_ctx *c, uint32_t b, AlnAcc &acc) {
    const uint32_t f = c->batch_first[b], l = c->batch_first[b + 1], cntp = l - f;
    if (cntp == 0) return SR_OK;
    std::vector<uint32_t> cnt(cntp);
    HIPCHK(hipMemcpy(cnt.data(), c->aa.cigar_cnt + f, (size_t)cntp * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(acc.score.data() + f, c->aa.score + f, (size_t)cntp * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(acc.isrev.data() + f, c->aa.is_reverse + f, (size_t)cntp, hipMemcpyDeviceToHost));
    const uint64_t arena_used = c->cigar_base[l] - c->cigar_base[f];
    std::vector<uint32_t> raw(arena_used + 1);
    HIPCHK(hipMemcpy(raw.data(), c->aa.cigar_ops, (arena_used + 1) * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < cntp; i++) {
        acc.cnt[f + i] = cnt[i];
        const uint32_t *src = raw.data() + (c->cigar_base[f + i] - c->cigar_base[f]);
        for (uint32_t j = 0; j < cnt[i]; j++) {
            const uint32_t op = src[j] & 15u, len = src[j] >> 4;
            // raw WFA2 alphabet -> reference alphabet (src/wfa.rs:25-31): M->'=', I<->D swapped
            const uint32_t o2 = op == SR_OP_M ? 0u : op == SR_OP_X ? 1u : op == SR_OP_I ? 3u : 2u;
            acc.ops.push_back((len << 4) | o2);
        }
    }
    return SR_OK;
}

static sr_alignments *make_alignments(const sr_ctx *c, const AlnAcc &acc) {
    const size_t np = c->pair_q.size();
    sr_alignments *a = (sr_alignments *)calloc(1, sizeof(sr_alignments));
    a->n = np;
    const size_t m = np ? np : 1;
    a->query_idx = (uint32_t *)malloc(m * 4); a->target_idx = (uint32_t *)malloc(m * 4);
    a->is_reverse = (uint8_t *)malloc(m); a->score = (int32_t *)malloc(m * 4);
    a->query_start = (uint64_t *)malloc(m * 8); a->query_end = (uint64_t *)malloc(m * 8);
    a->target_start = (uint64_t *)malloc(m * 8); a->target_end = (uint64_t *)malloc(m * 8);
    a->cigar_off = (uint64_t *)malloc((np + 1) * 8);
    a->cigar_ops = (uint32_t *)malloc((acc.ops.size() ? acc.ops.size() : 1) * 4);
    memcpy(a->cigar_ops, acc.ops.data(), acc.ops.size() * 4);
    uint64_t w = 0;
    for (size_t i = 0; i < np; i++) {
        a->query_idx[i] = c->pair_q[i]; a->target_idx[i] = c->pair_t[i];
        a->is_reverse[i] = acc.isrev[i]; a->score[i] = acc.score[i];
        a->query_start[i] = 0; a->query_end[i] = c->len[c->pair_q[i]];     // allwave aligns full sequences (seqrush.rs:743-753)
        a->target_start[i] = 0; a->target_end[i] = c->len[c->pair_t[i]];
        a->cigar_off[i] = w;
        w += acc.cnt[i];
    }
    a->cigar_off[np] = w;
    return a;
}

extern "C" int sr_ctx_alignments(sr_ctx *c, sr_alignments **out) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    if (c->from_paf) return fail(SR_ERR_INVALID, "context was loaded from a PAF file: it holds no device alignments");
    if (c->batch_first.size() > 2 || !c->aligned_batch_valid)
        return fail(SR_ERR_UNSUPPORTED, "the CIGARs of this context are not resident (several batches, or no alignment pass yet): use sr_align_all");
    int r = sr_ctx_sync(c);
    if (r) return r;
    const size_t np = c->pair_q.size();
    AlnAcc acc;
    acc.cnt.assign(np + 1, 0); acc.score.assign(np + 1, 0); acc.isrev.assign(np + 1, 0);
    if ((r = collect_batch(c, 0, acc))) return r;
    *out = make_alignments(c, acc);
    return SR_OK;
}

// per-pair results that stay resident for every batch: score, strand flag, CIGAR op count (arrays of num_pairs)
extern "C" int sr_ctx_pair_results(sr_ctx *c, int32_t *score, uint8_t *is_reverse, uint32_t *cigar_ops) {
    if (!c || !c->loaded || c->from_paf) return fail(SR_ERR_INVALID, "context holds no device alignments");
    int r = sr_ctx_sync(c);
    if (r) return r;
    const size_t np = c->pair_q.size();
    if (!np) return SR_OK;
    if (score) HIPCHK(hipMemcpy(score, c->aa.score, np * 4, hipMemcpyDeviceToHost));
    if (is_reverse) HIPCHK(hipMemcpy(is_reverse, c->aa.is_reverse, np, hipMemcpyDeviceToHost));
    if (cigar_ops) HIPCHK(hipMemcpy(cigar_ops, c->aa.cigar_cnt, np * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

extern "C" size_t sr_alignment_cigar(const sr_alignments *a, uint64_t i, char *buf, size_t cap) {
    static const char opc[4] = {'=', 'X', 'I', 'D'};
    size_t need = 0;
    if (!a || i >= a->n) return 0;
    for (uint64_t j = a->cigar_off[i]; j < a->cigar_off[i + 1]; j++) {
        char tmp[24];
        int l = snprintf(tmp, sizeof(tmp), "%u%c", a->cigar_ops[j] >> 4, opc[a->cigar_ops[j] & 3u]);
        if (buf && need + (size_t)l < cap) memcpy(buf + need, tmp, (size_t)l);
        need += (size_t)l;
    }
    if (buf && cap) buf[std::min(need, cap - 1)] = 0;
    return need;
}

// one batch at a time: align, copy its CIGARs to the host, (optionally) unite, next batch
static int align_all_impl(sr_ctx *c, bool unite, sr_alignments **out) {
    int r = begin_pass(c, true);
    if (r) return r;
    c->ev_used[1] = 0;
    const size_t np = c->pair_q.size();
    AlnAcc acc;
    acc.cnt.assign(np + 1, 0); acc.score.assign(np + 1, 0); acc.isrev.assign(np + 1, 0);
    const uint32_t nbatch = (uint32_t)c->batch_first.size() - 1;
    for (uint32_t b = 0; b < nbatch; b++) {
        if ((r = enqueue_align_batch(c, b))) return r;
        if (unite && (r = enqueue_unite_batch(c, b))) return r;
        if ((r = sr_ctx_sync(c))) return r;
        if ((r = collect_batch(c, b, acc))) return r;
    }
    c->aligned_batch_valid = nbatch == 1;
    *out = make_alignments(c, acc);
    return SR_OK;
}

extern "C" int sr_ctx_align_all(sr_ctx *c, int unite, sr_alignments **out) {
    if (!c || !c->loaded || !out) return fail(SR_ERR_INVALID, "context not loaded");
    if (c->from_paf) return fail(SR_ERR_INVALID, "context was loaded from a PAF file: there is no alignment stage");
    return align_all_impl(c, unite != 0, out);
}

extern "C" int sr_align_all(const sr_seqset *seqs, const sr_params *p, sr_alignments **out) {
    if (!seqs || !p || !out) return fail(SR_ERR_INVALID, "null argument");
    sr_ctx *c = nullptr;
    int r = sr_ctx_create(p->device, &c);
    if (r) return r;
    if (!(r = sr_ctx_load(c, seqs, p))) r = align_all_impl(c, false, out);
    std::string keep = g_err;
    sr_ctx_destroy(c);
    g_err = keep;
    return r;
}

extern "C" int sr_align_and_unite(const sr_seqset *seqs, const sr_params *p, uint64_t *parent_out) {
    if (!seqs || !p || !parent_out) return fail(SR_ERR_INVALID, "null argument");
    sr_ctx *c = nullptr;
    int r = sr_ctx_create(p->device, &c);
    if (r) return r;
    if (!(r = sr_ctx_load(c, seqs, p)) && !(r = sr_ctx_run(c)) && !(r = sr_ctx_sync(c))) {
        if (p->canonical_labels) r = sr_ctx_download_labels(c, parent_out);
        else r = sr_ctx_download_uf(c, parent_out);
        if (!r) r = sr_ctx_sync(c);
    }
    std::string keep = g_err;
    sr_ctx_destroy(c);
    g_err = keep;
    return r;
}

// ------------------------------------------------------------------ PAF input (seam 3, `seqrush -p`)
// SeqRush::align_and_unite_from_paf (src/seqrush.rs:510-609) + process_alignment (:1134-1481): every PAF
// record with a cg:Z: tag is replayed.  The host walks the CIGAR once and compares the bases of M / = ops
// (the reference does, :1268-1330), producing exact-match / mismatch runs in the device op alphabet; runs of
// consecutive M/= ops merge, so `len >= k` sees the same run lengths.  The records then go through the same
// sr_unite_kernel as device-made alignments, starting at (query_start, target_start); for strand '-' the
// query offset is in reverse-complement space like in the reference (:1210, 1165).
static inline uint8_t paf_query_base(const uint8_t *q, uint64_t qlen, bool rc, uint64_t pos) {   // :1162-1176
    return rc ? comp_base(q[qlen - 1 - pos]) : q[pos];
}

extern "C" int sr_ctx_load_paf(sr_ctx *c, const sr_seqset *seqs, const sr_params *p, const char *paf_path) {
    if (!c || !seqs || !p || !paf_path) return fail(SR_ERR_INVALID, "null argument");
    if (seqs->n == 0) return fail(SR_ERR_INVALID, "no sequences");
    if (!seqs->names) return fail(SR_ERR_INVALID, "PAF input needs sequence names");
    if (p->shard_count == 0 || p->shard_rank >= p->shard_count) return fail(SR_ERR_INVALID, "bad shard");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    free_dev(c);
    c->prm = *p;
    const uint32_t n = seqs->n;
    c->n = n;
    c->len.assign(n, 0); c->goff.assign(n, 0);
    std::unordered_map<std::string, uint32_t> by_name;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t L = seqs->offsets[i + 1] - seqs->offsets[i];
        std::string nm = seqs->names[i] ? seqs->names[i] : "";
        if (L == 0) return fail(SR_ERR_EMPTY_SEQ, "Empty sequences are not allowed: sequence '" + nm + "' has length 0");
        if (L > 0x7fff0000ULL) return fail(SR_ERR_UNSUPPORTED, "sequence too long");
        c->len[i] = (uint32_t)L; c->goff[i] = seqs->offsets[i];
        by_name[nm] = i;                                   // HashMap collect: a later duplicate id wins (:517-522)
    }
    c->total_len = seqs->offsets[n];
    c->uf_size = (c->total_len << 1) + 2;
    FILE *f = fopen(paf_path, "r");
    if (!f) return fail(SR_ERR_IO, std::string("Failed to open PAF file ") + paf_path);
    std::vector<uint32_t> ops, qs, ts;
    std::vector<uint8_t> isrev;
    c->pair_q.clear(); c->pair_t.clear();
    c->cigar_base.assign(1, 0);
    std::string line;
    std::vector<char> buf(1 << 16);
    uint64_t recno = 0;
    int rc_err = SR_OK;
    auto getline_all = [&](std::string &out) -> bool {
        out.clear();
        while (fgets(buf.data(), (int)buf.size(), f)) {
            out += buf.data();
            if (!out.empty() && out.back() == '\n') { out.pop_back(); if (!out.empty() && out.back() == '\r') out.pop_back(); return true; }
        }
        return !out.empty();
    };
    auto to_u64 = [&](const std::string &t, uint64_t &v) -> bool {
        if (t.empty()) return false;
        v = 0;
        for (char ch : t) { if (ch < '0' || ch > '9') return false; v = v * 10 + (uint64_t)(ch - '0'); }
        return true;
    };
    while (getline_all(line)) {
        if (line.empty()) continue;                                              // :529-531
        std::vector<std::string> fld;
        size_t b = 0;
        for (;;) { size_t e = line.find('\t', b); fld.push_back(line.substr(b, e == std::string::npos ? e : e - b)); if (e == std::string::npos) break; b = e + 1; }
        if (fld.size() < 12) continue;                                           // warning + skip (:534-537)
        uint64_t ql, qb, qe, tl, tb, te;
        if (!to_u64(fld[1], ql) || !to_u64(fld[2], qb) || !to_u64(fld[3], qe) || !to_u64(fld[6], tl) ||
            !to_u64(fld[7], tb) || !to_u64(fld[8], te)) {                        // .parse().unwrap() panics (:541-548)
            rc_err = fail(SR_ERR_INVALID, "PAF line with a non-numeric coordinate: " + line.substr(0, 80));
            break;
        }
        std::string cigar;
        for (size_t i = 12; i < fld.size(); i++) if (fld[i].compare(0, 5, "cg:Z:") == 0) { cigar = fld[i].substr(5); break; }
        auto iq = by_name.find(fld[0]), it = by_name.find(fld[5]);
        if (iq == by_name.end() || it == by_name.end()) continue;                // warning + skip (:570-576)
        const uint64_t my = recno++;
        if (my % p->shard_count != p->shard_rank) continue;
        const uint32_t q = iq->second, t = it->second;
        const bool rc = fld[4] == "-";
        if (qb > 0xffffffffULL || tb > 0xffffffffULL) { rc_err = fail(SR_ERR_INVALID, "PAF start coordinate out of range"); break; }
        const uint8_t *Q = seqs->bases + seqs->offsets[q], *T = seqs->bases + seqs->offsets[t];
        const uint64_t len1 = c->len[q], len2 = c->len[t];
        uint64_t pos1 = qb, pos2 = tb, count = 0;
        const size_t first = ops.size();
        auto push = [&](int op, uint64_t len) {
            while (len > 0) {
                const uint64_t chunk = std::min<uint64_t>(len, (1u << 27));
                if (ops.size() > first && (ops.back() & 15u) == (uint32_t)op && (ops.back() >> 4) + chunk < (1u << 28)) ops.back() += (uint32_t)chunk << 4;
                else ops.push_back(((uint32_t)chunk << 4) | (uint32_t)op);
                len -= chunk;
            }
        };
        bool bad = false;
        for (char ch : cigar) {
            if (ch >= '0' && ch <= '9') { count = count * 10 + (uint64_t)(ch - '0'); continue; }
            if (count == 0) count = 1;                                           // :1251-1253
            if (ch == 'M' || ch == '=') {
                for (uint64_t k = 0; k < count; k++) {
                    if (pos1 + k < len1 && pos2 + k < len2) {                    // :1268
                        if (paf_query_base(Q, len1, rc, pos1 + k) == T[pos2 + k]) push(SR_OP_M, 1);
                        else push(SR_OP_X, 1);
                    } else push(SR_OP_X, 1);    // past an end: never a match again (positions only grow); keeps offsets
                }
                pos1 += count; pos2 += count;
            } else if (ch == 'X') { push(SR_OP_X, count); pos1 += count; pos2 += count; }
            else if (ch == 'I') { push(SR_OP_D, count); pos1 += count; }         // PAF I = query only = raw D (:1420-1426)
            else if (ch == 'D') { push(SR_OP_I, count); pos2 += count; }         // PAF D = target only = raw I (:1427-1433)
            else { /* other ops: the reference ends the run and moves nothing (:1358-1440) */
                push(SR_OP_X, 0);
                if (ops.size() > first && (ops.back() & 15u) == SR_OP_M) ops.push_back((0u << 4) | SR_OP_X);
            }
            count = 0;
            if (pos1 > 0xffffffffULL || pos2 > 0xffffffffULL) { bad = true; break; }
        }
        if (bad) { rc_err = fail(SR_ERR_INVALID, "PAF CIGAR longer than 2^32 bases"); break; }
        c->pair_q.push_back(q); c->pair_t.push_back(t); isrev.push_back(rc ? 1 : 0);
        qs.push_back((uint32_t)qb); ts.push_back((uint32_t)tb);
        c->cigar_base.push_back(ops.size());
    }
    fclose(f);
    if (rc_err) return rc_err;
    const uint32_t np = (uint32_t)c->pair_q.size();
    std::vector<uint32_t> cnt(np ? np : 1, 0);
    for (uint32_t i = 0; i < np; i++) cnt[i] = (uint32_t)(c->cigar_base[i + 1] - c->cigar_base[i]);
    c->dp_cells = 0;
    // ---- device buffers
    int r;
    void *d;
    SrAlignArgs &a = c->aa;
    memset(&a, 0, sizeof(a));
#define DEV_UP(dst, T, hostvec)                                                               \
    do {                                                                                      \
        std::vector<T> tmp_ = (hostvec);                                                      \
        if (tmp_.empty()) tmp_.push_back(T());                                                \
        if ((r = dev_alloc(c, &d, tmp_.size() * sizeof(T)))) return r;                        \
        HIPCHK(hipMemcpy(d, tmp_.data(), tmp_.size() * sizeof(T), hipMemcpyHostToDevice));    \
        dst = (T *)d;                                                                         \
    } while (0)
    uint32_t *d_pq, *d_pt, *d_len, *d_ops, *d_cnt, *d_qs, *d_ts; uint64_t *d_goff, *d_cbase; uint8_t *d_rev; int32_t *d_score;
    DEV_UP(d_pq, uint32_t, c->pair_q); DEV_UP(d_pt, uint32_t, c->pair_t); DEV_UP(d_len, uint32_t, c->len);
    DEV_UP(d_goff, uint64_t, c->goff); DEV_UP(d_cbase, uint64_t, c->cigar_base); DEV_UP(d_ops, uint32_t, ops);
    DEV_UP(d_cnt, uint32_t, cnt); DEV_UP(d_qs, uint32_t, qs); DEV_UP(d_ts, uint32_t, ts); DEV_UP(d_rev, uint8_t, isrev);
    DEV_UP(d_score, int32_t, std::vector<int32_t>(np ? np : 1, 0));
    DEV_UP(c->d_max_score, int32_t, std::vector<int32_t>(np ? np : 1, INT_MAX));   // no divergence filter on PAF input
#undef DEV_UP
    if ((r = dev_alloc(c, &d, sizeof(uint32_t)))) return r; c->d_queue = (uint32_t *)d;
    if ((r = dev_alloc(c, &d, SR_NCOUNTERS * sizeof(unsigned long long)))) return r; c->d_counters = (unsigned long long *)d;
    if ((r = dev_alloc(c, &d, sizeof(int)))) return r; c->d_error = (int *)d;
    if ((r = dev_alloc(c, &d, c->uf_size * 8))) return r; c->d_nodes = (unsigned long long *)d;
    if ((r = dev_alloc(c, &d, c->uf_size * 8))) return r; c->d_minarr = (unsigned long long *)d;
    if ((r = dev_alloc(c, &d, c->uf_size * 8))) return r; c->d_labels = (unsigned long long *)d;
    HIPCHK(hipMemsetAsync(c->d_counters, 0, SR_NCOUNTERS * sizeof(unsigned long long), c->stream));
    HIPCHK(hipMemsetAsync(c->d_error, 0, sizeof(int), c->stream));
    a.npairs = 0; a.counters = c->d_counters; a.error_flag = c->d_error;
    SrUniteArgs &u = c->ua;
    memset(&u, 0, sizeof(u));
    u.pair_q = d_pq; u.pair_t = d_pt; u.npairs = np; u.seqlen = d_len; u.seq_goff = d_goff;
    u.q_start = d_qs; u.t_start = d_ts;
    u.is_reverse = d_rev; u.score = d_score; u.max_score = c->d_max_score;
    u.cigar_ops = d_ops; u.cigar_base = d_cbase; u.cigar_cnt = d_cnt;
    u.min_match_len = p->min_match_len; u.nodes = c->d_nodes; u.uf_size = c->uf_size;
    u.counters = c->d_counters; u.error_flag = c->d_error;
    c->batch_first.assign(1, 0); c->batch_first.push_back(np);          // PAF records: one batch
    c->d_cbase = d_cbase; c->total_pairs_all_ranks = recno;
    if (srk_uf_init(c->d_nodes, c->total_len, c->uf_size, c->stream)) return fail(SR_ERR_HIP, "uf init launch failed");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->loaded = true; c->from_paf = true;
    return SR_OK;
}

extern "C" int sr_unite_paf(const sr_seqset *seqs, const sr_params *p, const char *paf_path, uint64_t *parent_out) {
    if (!seqs || !p || !paf_path || !parent_out) return fail(SR_ERR_INVALID, "null argument");
    sr_ctx *c = nullptr;
    int r = sr_ctx_create(p->device, &c);
    if (r) return r;
    if (!(r = sr_ctx_load_paf(c, seqs, p, paf_path)) && !(r = sr_ctx_run(c)) && !(r = sr_ctx_sync(c))) {
        if (p->canonical_labels) r = sr_ctx_download_labels(c, parent_out);
        else r = sr_ctx_download_uf(c, parent_out);
        if (!r) r = sr_ctx_sync(c);
    }
    std::string keep = g_err;
    sr_ctx_destroy(c);
    g_err = keep;
    return r;
}


// UFRush::find (read-only walk) / same, uf_rush lib.rs:112-133, 72-84
extern "C" uint64_t sr_uf_find(const uint64_t *nodes, uint64_t n, uint64_t x) {
    const uint64_t mask = 0x03FFFFFFFFFFFFFFULL;
    if (x >= n) return UINT64_MAX;
    uint64_t guard = 0;
    while ((nodes[x] & mask) != x && guard++ < n) x = nodes[x] & mask;
    return x;
}
extern "C" int sr_uf_same(const uint64_t *nodes, uint64_t n, uint64_t x, uint64_t y) {
    return sr_uf_find(nodes, n, x) == sr_uf_find(nodes, n, y);
}

// Host-side uf_rush over a node array (same packing parent | rank << 58, path halving, union by rank, tie rule "larger
// index wins" as uf_rush lib.rs:112-208; one thread, so the CAS loops collapse to plain stores).  For hosts that merge
// per-GPU forests or replay unites themselves without a device: SURVEY 8(e)'s merge and 8(b)'s Seam 2 consumers.
static const uint64_t UFH_MASK = 0x03FFFFFFFFFFFFFFULL;
static inline uint64_t ufh_find(uint64_t *nodes, uint64_t x) {          // UFRush::find, lib.rs:112-133
    uint64_t xn = nodes[x];
    while (x != (xn & UFH_MASK)) {
        const uint64_t par = xn & UFH_MASK, pp = nodes[par] & UFH_MASK;
        nodes[x] = pp | (xn & ~UFH_MASK);                                // path halving: parent := grandparent, rank bits kept
        x = pp; xn = nodes[x];
    }
    return x;
}
static inline bool ufh_unite(uint64_t *nodes, uint64_t x, uint64_t y) { // UFRush::unite, lib.rs:159-208
    uint64_t xr = ufh_find(nodes, x), yr = ufh_find(nodes, y);
    if (xr == yr) return false;
    uint64_t xk = nodes[xr] >> 58, yk = nodes[yr] >> 58;
    if (xk > yk || (xk == yk && xr > yr)) { std::swap(xr, yr); std::swap(xk, yk); }
    nodes[xr] = yr | (xk << 58);
    if (xk == yk) nodes[yr] = yr | ((yk + 1) << 58);
    return true;
}
// SeqRush::new (src/seqrush.rs:308-336): 2N+2 nodes, unite(fwd(i), rev(i)) for every offset i, in order
extern "C" int sr_uf_init_host(uint64_t *nodes, uint64_t n, uint64_t total_len) {
    if (!nodes || n < 2 * total_len + 2) return fail(SR_ERR_INVALID, "node array shorter than 2 * total_len + 2");
    for (uint64_t i = 0; i < n; i++) nodes[i] = i;
    for (uint64_t i = 0; i < total_len; i++) ufh_unite(nodes, 2 * i, 2 * i + 1);
    return SR_OK;
}
extern "C" int sr_uf_unite_host(uint64_t *nodes, uint64_t n, uint64_t x, uint64_t y) {
    if (!nodes || x >= n || y >= n) return fail(SR_ERR_INVALID, "union-find index out of range");   // uf_rush panics (lib.rs:113)
    return ufh_unite(nodes, x, y) ? 1 : 0;
}
// SURVEY 8(e) merge on the host: replay unite(i, labels_g[i]) for `count` label arrays of n entries laid out back to back
extern "C" int sr_uf_merge_labels_host(uint64_t *nodes, uint64_t n, const uint64_t *labels, uint32_t count) {
    if (!nodes || !labels) return fail(SR_ERR_INVALID, "null argument");
    for (uint32_t g = 0; g < count; g++) {
        const uint64_t *lab = labels + (uint64_t)g * n;
        for (uint64_t i = 0; i < n; i++) {
            if (lab[i] >= n) return fail(SR_ERR_INVALID, "label out of range");
            if (lab[i] != i) ufh_unite(nodes, i, lab[i]);
        }
    }
    return SR_OK;
}
// canonical labels of a node array: the minimum element of each set (what sr_ctx_download_labels returns)
extern "C" int sr_uf_canonical_labels_host(const uint64_t *nodes, uint64_t n, uint64_t *labels_out) {
    if (!nodes || !labels_out) return fail(SR_ERR_INVALID, "null argument");
    std::vector<uint64_t> root(n), mn(n, UINT64_MAX);
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t r = sr_uf_find(nodes, n, i);
        if (r >= n) return fail(SR_ERR_INVALID, "node array is not a uf_rush forest");
        root[i] = r;
        if (mn[r] == UINT64_MAX) mn[r] = i;                              // ascending i: the first one seen is the minimum
    }
    for (uint64_t i = 0; i < n; i++) labels_out[i] = mn[root[i]];
    return SR_OK;
}

// ------------------------------------------------------------------ PAF (seam 3)
extern "C" int sr_write_paf(const sr_alignments *a, const sr_seqset *seqs, const char *path) {
    if (!a || !seqs || !path || !seqs->names) return fail(SR_ERR_INVALID, "null argument");
    FILE *f = fopen(path, "w");
    if (!f) return fail(SR_ERR_IO, std::string("cannot open ") + path);
    std::vector<char> buf;
    for (uint64_t i = 0; i < a->n; i++) {
        const uint32_t q = a->query_idx[i], t = a->target_idx[i];
        uint64_t matches = 0, alen = 0;
        for (uint64_t j = a->cigar_off[i]; j < a->cigar_off[i + 1]; j++) {
            const uint32_t len = a->cigar_ops[j] >> 4;
            if ((a->cigar_ops[j] & 3u) == 0) matches += len;
            alen += len;
        }
        const size_t need = sr_alignment_cigar(a, i, nullptr, 0);
        buf.resize(need + 1);
        sr_alignment_cigar(a, i, buf.data(), need + 1);
        // 12 mandatory PAF columns + cg:Z: (parsed by seqrush.rs:536-559)
        fprintf(f, "%s\t%llu\t%llu\t%llu\t%c\t%s\t%llu\t%llu\t%llu\t%llu\t%llu\t255\tAS:i:%d\tcg:Z:%s\n",
                seqs->names[q], (unsigned long long)(seqs->offsets[q + 1] - seqs->offsets[q]),
                (unsigned long long)a->query_start[i], (unsigned long long)a->query_end[i],
                a->is_reverse[i] ? '-' : '+', seqs->names[t],
                (unsigned long long)(seqs->offsets[t + 1] - seqs->offsets[t]),
                (unsigned long long)a->target_start[i], (unsigned long long)a->target_end[i],
                (unsigned long long)matches, (unsigned long long)alen, a->score[i], buf.data());
    }
    fclose(f);
    return SR_OK;
}

// ------------------------------------------------------------------ GFA (A9)
// O(N) formulation of build_bidirected_graph_with_options
// (bidirected_builder.rs:17-289) for a quiescent UF given as canonical labels:
// node ids in first-encounter order over sequences / positions (:29-41,
// :154-157), node base = base at offset(label) (:176-182), step reversed iff
// node base and sequence base are complementary (:190-203), edges deduplicated
// against themselves and their complement, first orientation kept
// (bidirected_ops.rs:813-825); optional compact() + renumber (sr_compact.cpp); write_gfa layout
// bidirected_ops.rs:880-925.
struct PairHash {
    size_t operator()(const std::pair<uint64_t, uint64_t> &p) const {
        return (size_t)splitmix64(p.first * 0x9e3779b97f4a7c15ULL ^ p.second);
    }
};

static char *finish_gfa(SrGraph &g, const sr_seqset *seqs, int compact, uint64_t *n_nodes, uint64_t *n_edges) {
    if (compact) { sr_graph_compact(g); sr_graph_renumber(g); }     // src/bidirected_gfa_writer.rs:39-51
    return sr_graph_format_gfa(g, seqs->names, n_nodes, n_edges);
}

extern "C" int sr_build_gfa_opts(const sr_seqset *seqs, const uint64_t *labels, int compact, char **gfa,
                                 uint64_t *n_nodes, uint64_t *n_edges) {
    if (!seqs || !labels || !gfa || !seqs->names) return fail(SR_ERR_INVALID, "null argument");
    const uint64_t N = seqs->offsets[seqs->n];
    if (N >= 0x7fffffffULL) return fail(SR_ERR_UNSUPPORTED, "graph induction supports < 2^31 bases");
    const uint64_t ufn = 2 * N + 2;
    std::vector<uint32_t> node_of(ufn, 0);
    SrGraph g;
    g.node_seq.assign(1, std::string()); g.node_alive.assign(1, 0);
    g.steps.resize(N); g.path_off.assign(1, 0);
    uint32_t next_id = 1;
    for (uint32_t s = 0; s < seqs->n; s++) {
        for (uint64_t p = seqs->offsets[s]; p < seqs->offsets[s + 1]; p++) {
            const uint64_t lf = labels[p << 1], lr = labels[(p << 1) | 1];
            if (lf >= ufn || lr >= ufn) return fail(SR_ERR_INVALID, "label out of range");
            const uint64_t rep = node_of[lf] ? lf : (node_of[lr] ? lr : lf);
            uint32_t id = node_of[rep];
            if (!id) {
                id = next_id++;
                node_of[rep] = id;
                const uint64_t off = rep >> 1;
                g.node_seq.push_back(std::string(1, (char)(off < N ? seqs->bases[off] : seqs->bases[p])));
                g.node_alive.push_back(1);
            }
            const uint8_t nb = (uint8_t)toupper((unsigned char)g.node_seq[id][0]), eb = (uint8_t)toupper(seqs->bases[p]);
            const bool rev = (nb == 'A' && eb == 'T') || (nb == 'T' && eb == 'A') ||
                             (nb == 'C' && eb == 'G') || (nb == 'G' && eb == 'C');
            g.steps[p] = (id << 1) | (rev ? 1u : 0u);
        }
        g.path_off.push_back(seqs->offsets[s + 1]);
    }
    std::unordered_set<std::pair<uint64_t, uint64_t>, PairHash> eset;
    eset.reserve(N);
    for (uint32_t s = 0; s < seqs->n; s++)
        for (uint64_t p = seqs->offsets[s]; p + 1 < seqs->offsets[s + 1]; p++) {
            const uint32_t from = g.steps[p], to = g.steps[p + 1];
            if (eset.count({from, to}) || eset.count({to ^ 1u, from ^ 1u})) continue;
            eset.insert({from, to});
            g.edges.push_back({from, to});
        }
    *gfa = finish_gfa(g, seqs, compact, n_nodes, n_edges);
    return SR_OK;
}
extern "C" int sr_build_gfa(const sr_seqset *seqs, const uint64_t *labels, char **gfa, uint64_t *n_nodes, uint64_t *n_edges) {
    return sr_build_gfa_opts(seqs, labels, 0, gfa, n_nodes, n_edges);
}
// The reference's own root rule (src/bidirected_builder.rs:46-48, 176-182): a node takes the base at the offset of its
// component's union-find ROOT (union_find.find(pos)), whichever element the unite order made the root -- not the minimum
// Pos the canonical entry points above use.  Given the raw uf_rush node array (sr_align_and_unite with canonical_labels
// = 0, sr_ctx_download_uf, or an array a host built by replaying unites in its own fixed order through uf_rush /
// sr_uf_unite_host), the representative of every element is its root and the induction is otherwise the same: GFA
// byte equality with a reference run whose unites happened in that order, node orientation included.
extern "C" int sr_build_gfa_from_nodes(const sr_seqset *seqs, const uint64_t *nodes, int compact, char **gfa,
                                       uint64_t *n_nodes, uint64_t *n_edges) {
    if (!seqs || !nodes || !gfa || !seqs->names) return fail(SR_ERR_INVALID, "null argument");
    const uint64_t N = seqs->offsets[seqs->n], ufn = 2 * N + 2;
    std::vector<uint64_t> root(ufn);
    for (uint64_t i = 0; i < ufn; i++) {
        root[i] = sr_uf_find(nodes, ufn, i);
        if (root[i] >= ufn) return fail(SR_ERR_INVALID, "node array is not a uf_rush forest");
    }
    return sr_build_gfa_opts(seqs, root.data(), compact, gfa, n_nodes, n_edges);
}

// graph induction on the device from the context's union-find (sr_graph.hip); same
// text as sr_build_gfa() on the downloaded canonical labels.  which = 3 of sr_ctx_kernel_ms times it.
extern "C" int sr_ctx_build_gfa_opts(sr_ctx *c, const sr_seqset *seqs, int compact, char **gfa, uint64_t *n_nodes, uint64_t *n_edges) {
    if (!c || !c->loaded) return fail(SR_ERR_INVALID, "context not loaded");
    if (!seqs || !gfa || !seqs->names) return fail(SR_ERR_INVALID, "null argument");
    const uint64_t N = seqs->offsets[seqs->n];
    if (seqs->n != c->n || N != c->total_len) return fail(SR_ERR_INVALID, "sequence set differs from the loaded one");
    if (N >= 0x7fffffffULL) return fail(SR_ERR_UNSUPPORTED, "graph induction on device supports < 2^31 bases");
    HIPCHK(hipSetDevice(c->device));
    int r = sr_ctx_labels_device(c, (uint64_t *)c->d_labels);
    if (r) return r;
    std::vector<uint8_t> islast(N, 0);
    for (uint32_t s = 0; s < seqs->n; s++) islast[seqs->offsets[s + 1] - 1] = 1;
    uint64_t hcap = 64;
    while (hcap < 2 * N + 16) hcap <<= 1;
    const uint64_t ntiles = (N + 1023) / 1024 + 1;
    struct Tmp { std::vector<void *> v; ~Tmp() { for (void *p : v) (void)hipFree(p); } } tmp;
    auto dalloc = [&](size_t bytes) -> void * { void *p = nullptr; if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr; tmp.v.push_back(p); return p; };
    uint8_t *d_bases = c->d_bases, *d_islast = (uint8_t *)dalloc(N), *d_nbase = (uint8_t *)dalloc(N);
    bool own_bases = false;
    if (!d_bases) { d_bases = (uint8_t *)dalloc(N); own_bases = true; }        // (PAF contexts keep no byte copy)
    uint32_t *d_flag = (uint32_t *)dalloc(N * 4), *d_nid = (uint32_t *)dalloc(N * 4), *d_steps = (uint32_t *)dalloc((N + 1) * 4);
    uint32_t *d_eslot = (uint32_t *)dalloc(N * 4), *d_hvals = (uint32_t *)dalloc(hcap * 4), *d_tiles = (uint32_t *)dalloc(ntiles * 4);
    uint32_t *d_counts = (uint32_t *)dalloc(8);
    unsigned long long *d_hkeys = (unsigned long long *)dalloc(hcap * 8), *d_edges = (unsigned long long *)dalloc(N * 8);
    if (!d_bases || !d_islast || !d_nbase || !d_flag || !d_nid || !d_steps || !d_eslot || !d_hvals || !d_tiles || !d_counts ||
        !d_hkeys || !d_edges) return fail(SR_ERR_NOMEM, "not enough device memory for graph induction");
    if (own_bases) HIPCHK(hipMemcpyAsync(d_bases, seqs->bases, N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_islast, islast.data(), N, hipMemcpyHostToDevice, c->stream));
    hipEvent_t *ge0, *ge1;
    if ((r = ev_get(c, 3, 0, &ge0, &ge1))) return r;
    HIPCHK(hipEventRecord(*ge0, c->stream));
    // d_minarr (uf_size entries) is free again once the labels exist: first-position table
    if (srk_graph_induce(c->d_labels, d_bases, d_islast, N, c->uf_size, c->d_minarr, d_flag, d_nid, d_steps, d_nbase, d_hkeys,
                         d_hvals, hcap, d_eslot, d_edges, d_tiles, d_counts, c->d_error, c->stream))
        return fail(SR_ERR_HIP, "graph induction launch failed");
    HIPCHK(hipEventRecord(*ge1, c->stream));
    c->ev_used[3] = 1;
    if ((r = sr_ctx_sync(c))) return r;
    uint32_t counts[2] = {0, 0};
    HIPCHK(hipMemcpy(counts, d_counts, 8, hipMemcpyDeviceToHost));
    std::vector<uint8_t> nbase(counts[0] ? counts[0] : 1);
    std::vector<unsigned long long> edges(counts[1] ? counts[1] : 1);
    SrGraph g;
    g.steps.resize(N);
    HIPCHK(hipMemcpy(nbase.data(), d_nbase, counts[0], hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(g.steps.data(), d_steps, N * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(edges.data(), d_edges, (size_t)counts[1] * 8, hipMemcpyDeviceToHost));
    g.node_seq.assign((size_t)counts[0] + 1, std::string()); g.node_alive.assign((size_t)counts[0] + 1, 1);
    g.node_alive[0] = 0;
    for (uint32_t i = 0; i < counts[0]; i++) g.node_seq[i + 1].assign(1, (char)nbase[i]);
    g.path_off.assign(1, 0);
    for (uint32_t s = 0; s < seqs->n; s++) g.path_off.push_back(seqs->offsets[s + 1]);
    g.edges.resize(counts[1]);
    for (uint32_t i = 0; i < counts[1]; i++) g.edges[i] = {(uint32_t)(edges[i] >> 32), (uint32_t)(edges[i] & 0xffffffffULL)};
    *gfa = finish_gfa(g, seqs, compact, n_nodes, n_edges);
    return SR_OK;
}
extern "C" int sr_ctx_build_gfa(sr_ctx *c, const sr_seqset *seqs, char **gfa, uint64_t *n_nodes, uint64_t *n_edges) {
    return sr_ctx_build_gfa_opts(c, seqs, 0, gfa, n_nodes, n_edges);
}
