/*
 * seqrush_amd.h -- C ABI of the MI355X-native seqrush hot path
 * (all-vs-all WFA2/biWFA alignment -> match-run extraction -> lock-free
 * bidirected union-find).  Plain pointers and sizes only; no torch types.
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the pangenome/seqrush checkout).  INTEGRATION.md shows the Rust
 * `extern "C"` binding a seqrush maintainer would add.
 *
 * Error model: every int-returning function returns 0 on success or a
 * negative sr_status; sr_last_error() gives a thread-local message.  Nothing
 * here falls back to a CPU path: without a HIP device every compute entry
 * point fails with SR_ERR_NO_DEVICE.
 */
#ifndef SEQRUSH_AMD_H
#define SEQRUSH_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_ABI_VERSION 2

typedef enum {
    SR_OK = 0,
    SR_ERR_INVALID = -1,      /* bad argument / parse error */
    SR_ERR_NO_DEVICE = -2,    /* no HIP device or HIP runtime failure at init */
    SR_ERR_HIP = -3,          /* a HIP call failed */
    SR_ERR_ALPHABET = -4,     /* (unused since ABI 2: every byte value is accepted, like the reference's raw-byte compare) */
    SR_ERR_EMPTY_SEQ = -5,    /* "Empty sequences are not allowed" seqrush.rs:310-317 */
    SR_ERR_UNSUPPORTED = -6,  /* parameter combination not implemented on device */
    SR_ERR_DEVICE_FAULT = -7, /* kernel reported an internal bound violation */
    SR_ERR_IO = -8,
    SR_ERR_NOMEM = -9
} sr_status;

/* -------- inputs --------------------------------------------------------
 * sr_seqset mirrors `&[AlignmentSequence]` (src/aligner.rs:5-9) /
 * `Vec<Sequence{id,data,offset}>` (src/seqrush.rs:272-277): concatenated
 * bases + offsets[n+1] (offset = running sum, seqrush.rs:1818) + names. */
typedef struct {
    uint32_t n;
    const uint8_t *bases;         /* offsets[n] bytes */
    const uint64_t *offsets;      /* n+1 entries, offsets[0] = 0 */
    const char *const *names;     /* n C strings (may be NULL for unite-only use) */
} sr_seqset;

/* lib_wfa2 MemoryMode as used by src/wfa.rs:57,65.  The reference always selects Ultralow (= biWFA); the device
 * keeps full wavefront history only for biWFA's base cases, so SR_MEM_HIGH is refused with SR_ERR_UNSUPPORTED. */
#define SR_MEM_HIGH 0
#define SR_MEM_ULTRALOW 3

/* SparsificationStrategy (grammar src/seqrush.rs:356-431) */
#define SR_SPARSE_NONE 0
#define SR_SPARSE_AUTO 1
#define SR_SPARSE_RANDOM 2
#define SR_SPARSE_CONNECTIVITY 3
#define SR_SPARSE_TREE 4

/* sr_params mirrors allwave::AlignmentParams (src/seqrush.rs:648-666) plus
 * the Args fields the hot path reads (-k, -d, -x; src/seqrush.rs:24-151). */
typedef struct {
    int32_t match_score;        /* must be 0: seqrush.rs:45; the trait impl's own 2,4,4,2,24,1 (allwave_impl.rs:15-23)
                                   is refused, allwave's conversion of a positive match score is not in the reference tree */
    int32_t mismatch_penalty;   /* -S default 0,5,8,2,24,1 (seqrush.rs:45) */
    int32_t gap_open1, gap_ext1;
    int32_t gap_open2, gap_ext2;        /* < 0 : single-piece affine */
    int32_t ori_match, ori_mismatch, ori_gap_open, ori_gap_ext; /* 0,1,1,1 (:49) */
    uint64_t min_match_len;     /* -k (:33), run united iff len >= k (:1311) */
    double max_divergence;      /* -d, < 0 = None */
    int32_t exclude_self;       /* reference passes false (:731) */
    int32_t memory_mode;        /* SR_MEM_ULTRALOW = reference (wfa.rs:57) */
    int32_t sparsify_kind;      /* SR_SPARSE_* (grammar seqrush.rs:356-431).  allwave's selection rules are absent from the
                                   reference tree: the definitions (sr_host.cpp "pair list", sr_sketch.hip) are this
                                   project's own and unpinned, random:F included */
    double sparsify_factor;     /* random:F / connectivity:P */
    uint64_t sparsify_seed;
    int32_t canonical_labels;   /* sr_align_and_unite: 1 = return min-Pos labels */
    int32_t device;             /* HIP device ordinal */
    /* pair shard for multi-GPU: this call handles shard `shard_rank` of `shard_count` of the (sparsified) row-major
     * ordered pair list.  The shards are cost-balanced: pairs sorted by |q|*|t| (self pairs: |q|), longest first, each
     * dealt to the rank with the least work so far (longest-processing-time-first; every rank computes the same
     * assignment; equal lengths degenerate to index mod shard_count -- sr_host.cpp shard_pairs) */
    uint32_t shard_rank, shard_count;
    /* tree:neighbor[,stranger[,random[,k-mer]]] (seqrush.rs:378-418; extract_tree_pairs_separated call :941-947) */
    uint32_t tree_k_nearest, tree_k_farthest;
    double tree_rand_frac;
    uint32_t tree_kmer;         /* default 16; 1..32 on the device */
} sr_params;

void sr_default_params(sr_params *p);
/* AlignmentScores::parse seqrush.rs:165-217 / parse_orientation :219-250 */
int sr_parse_scores(const char *s, sr_params *p);
int sr_parse_orientation_scores(const char *s, sr_params *p);
/* parse_sparsification seqrush.rs:356-431 */
int sr_parse_sparsification(const char *s, sr_params *p);

/* This rank's ordered pair list (host only, no device needed): what
 * AllPairIterator::with_options(.., exclude_self, .., sparsification) enumerates
 * (src/seqrush.rs:728-735), sharded for multi-GPU (equal sequence lengths assumed: sr_ctx_pairs gives the
 * cost-balanced shard of a loaded context).  tree: needs the sequences -> SR_ERR_UNSUPPORTED here.
 * Free both arrays with sr_free. */
int sr_pair_list(uint32_t n, const sr_params *p, uint32_t **q_out, uint32_t **t_out,
                 uint64_t *count);

/* -------- Seam 1: trait Aligner (src/aligner.rs:27-33) -----------------
 * sr_align_all == AllwaveAligner::align_sequences
 * (src/aligner/allwave_impl.rs:95-149): one call, all sequences, returns one
 * record per ordered pair (self pairs included unless exclude_self). */
typedef struct {
    uint64_t n;                 /* number of alignments */
    uint32_t *query_idx, *target_idx;
    uint8_t *is_reverse;        /* strand '-' <=> 1 (aligner.rs:18) */
    int32_t *score;
    uint64_t *query_start, *query_end, *target_start, *target_end;
    uint64_t *cigar_off;        /* n+1 offsets into cigar_ops */
    uint32_t *cigar_ops;        /* (len << 4) | op ; op: 0 '=' 1 'X' 2 'I' 3 'D'
                                   in the reference's converted alphabet
                                   (src/wfa.rs:25-31: I = query-only, D = target-only) */
} sr_alignments;

int sr_align_all(const sr_seqset *seqs, const sr_params *p, sr_alignments **out);
void sr_alignments_free(sr_alignments *a);
/* CIGAR string of alignment i ("159=1X75=..."), == cigar_bytes_to_string
 * (src/wfa.rs:9-38).  Returns needed length (excl. NUL); writes up to cap. */
size_t sr_alignment_cigar(const sr_alignments *a, uint64_t i, char *buf, size_t cap);

/* -------- Seam 2: fused production path ---------------------------------
 * == SeqRush::new (seqrush.rs:308-336) + align_and_unite_with_allwave
 * (seqrush.rs:611-757): alignment + process_alignment (:1134-1481) +
 * BidirectedUnionFind::unite_matching_region (bidirected_union_find.rs:60-98)
 * entirely on device.  parent_out has 2*N+2 entries: the uf_rush node array
 * (parent | rank << 58, uf_rush lib.rs:228-240) or, with canonical_labels,
 * the minimum Pos of each element's component. */
int sr_align_and_unite(const sr_seqset *seqs, const sr_params *p, uint64_t *parent_out);

/* uf_rush find/same over a returned node array (host side, read-only, no
 * path compression): UFRush::find lib.rs:112-133, same :72-84 */
uint64_t sr_uf_find(const uint64_t *nodes, uint64_t n, uint64_t x);
int sr_uf_same(const uint64_t *nodes, uint64_t n, uint64_t x, uint64_t y);
/* The same forest operations for hosts that build or merge node arrays WITHOUT a device (one thread, plain stores
 * instead of CAS; identical packing, path halving, union by rank and tie rule -- larger index wins -- as
 * uf_rush-0.2.1/src/lib.rs:112-208):
 *   sr_uf_init_host             SeqRush::new state (src/seqrush.rs:308-336) for total_len bases, n >= 2*total_len+2
 *   sr_uf_unite_host            UFRush::unite (lib.rs:159-208); returns 1 if two sets were joined, 0 if already one,
 *                               negative on an index out of range (the reference panics, lib.rs:113)
 *   sr_uf_merge_labels_host     SURVEY 8(e) merge: replay unite(i, labels_g[i]) for `count` gathered canonical label
 *                               arrays (n entries each, back to back) into `nodes` -- the host twin of
 *                               sr_ctx_merge_labels, for a Rust host that gathers per-GPU labels itself
 *   sr_uf_canonical_labels_host minimum element of every set (== sr_ctx_download_labels of that forest) */
int sr_uf_init_host(uint64_t *nodes, uint64_t n, uint64_t total_len);
int sr_uf_unite_host(uint64_t *nodes, uint64_t n, uint64_t x, uint64_t y);
int sr_uf_merge_labels_host(uint64_t *nodes, uint64_t n, const uint64_t *labels, uint32_t count);
int sr_uf_canonical_labels_host(const uint64_t *nodes, uint64_t n, uint64_t *labels_out);

/* -------- Seam 3: PAF interchange (seqrush.rs:510-609, 678-716) -------- */
int sr_write_paf(const sr_alignments *a, const sr_seqset *seqs, const char *path);

/* -------- resident-context API (what bench.py times) --------------------
 * Same path split so inputs can stay in HBM across calls. */
typedef struct sr_ctx sr_ctx;
int sr_ctx_create(int device, sr_ctx **out);
void sr_ctx_destroy(sr_ctx *c);
/* optional: launch on an externally owned hipStream_t */
int sr_ctx_set_stream(sr_ctx *c, void *hip_stream);
/* pack (2 / 4 / 8 bits per symbol by alphabet; forward + reverse complement) and upload; pair list (sparsified:
 * k-mer sketches on the device), cost-balanced shard, SeqRush::new UF init */
int sr_ctx_load(sr_ctx *c, const sr_seqset *seqs, const sr_params *p);
/* the same with an explicit ordered pair list instead of the enumeration: the per-pair call pattern of the
 * reference's iterative mode (AllPairIterator over a 2-sequence slice, src/seqrush.rs:984-1012) and of
 * seqrush_clean (src/seqrush_clean.rs:289-292) without re-uploading the sequences; sharded like sr_ctx_load */
int sr_ctx_load_pairs(sr_ctx *c, const sr_seqset *seqs, const sr_params *p, const uint32_t *query_idx,
                      const uint32_t *target_idx, uint64_t count);
/* this rank's pair list of the loaded context (after sparsification and sharding); free with sr_free */
int sr_ctx_pairs(const sr_ctx *c, uint32_t **q_out, uint32_t **t_out, uint64_t *count);
/* The CIGAR arena holds |q|+|t|+2 ops per pair; when the shard's pairs do not fit device memory at once they run in
 * batches that reuse it (1 for every BASELINE config up to C4; C5 on one GPU: several) */
uint32_t sr_ctx_num_batches(const sr_ctx *c);
/* JSON object describing the sizing of the last load (workgroups, ring / history / arena bytes ...) */
const char *sr_ctx_workspace_report(const sr_ctx *c);
/* reset the UF to the SeqRush::new state (seqrush.rs:324-328) */
int sr_ctx_reset_uf(sr_ctx *c);
/* enqueue alignment kernel(s) for this rank's pair shard (no host sync); single-batch contexts only */
int sr_ctx_align(sr_ctx *c);
/* enqueue match-run extraction + unite kernel (no host sync); single-batch contexts only */
int sr_ctx_unite(sr_ctx *c);
/* align + unite of the whole shard, batch after batch (no host sync): == sr_ctx_align + sr_ctx_unite when there
 * is one batch; for a PAF context: unite only */
int sr_ctx_run(sr_ctx *c);
/* Seam 1 on a loaded context: every alignment of the shard on the host (batches are copied out one after the
 * other); unite != 0 also unites each batch (--output-alignments without the reference's second pass) */
int sr_ctx_align_all(sr_ctx *c, int unite, sr_alignments **out);
/* per-pair results that stay resident across batches (arrays of sr_ctx_num_pairs; NULL = skip): score (-1 = failed),
 * strand flag, number of run-length CIGAR ops */
int sr_ctx_pair_results(sr_ctx *c, int32_t *score, uint8_t *is_reverse, uint32_t *cigar_ops);
int sr_ctx_sync(sr_ctx *c);     /* stream sync + device error flags */
/* results */
int sr_ctx_alignments(sr_ctx *c, sr_alignments **out);       /* after sync */
int sr_ctx_download_uf(sr_ctx *c, uint64_t *parent_out);     /* raw nodes */
uint64_t sr_ctx_uf_size(const sr_ctx *c);                    /* 2N+2 */
uint64_t sr_ctx_num_pairs(const sr_ctx *c);                  /* this shard */
uint64_t sr_ctx_dp_cells(const sr_ctx *c);   /* sum |q|*|t| over this shard */
/* multi-GPU merge: canonical min-Pos labels of the local forest
 * written to a DEVICE buffer of uf_size u64 (e.g. a torch tensor's data_ptr);
 * sr_ctx_merge_labels replays unite(i, labels[i]) for `count` gathered arrays
 * laid out back to back in DEVICE memory. */
int sr_ctx_labels_device(sr_ctx *c, uint64_t *dev_labels);
int sr_ctx_merge_labels(sr_ctx *c, const uint64_t *dev_labels, uint32_t count);
/* the same exchange with 32-bit labels (valid while 2N+2 < 2^32: half the all-gather bytes) */
int sr_ctx_labels_device_u32(sr_ctx *c, uint32_t *dev_labels);
int sr_ctx_merge_labels_u32(sr_ctx *c, const uint32_t *dev_labels, uint32_t count);
int sr_ctx_download_labels(sr_ctx *c, uint64_t *labels_out);
/* the merge with `count` label arrays resident on the HOST (uf_size entries each): hosts without a collective
 * library exchange the arrays through files (seqrush_mi355x --shard R/N --labels-out / --labels-in) */
int sr_ctx_merge_labels_host(sr_ctx *c, const uint64_t *labels, uint32_t count);
/* Seam 3, input side (`seqrush -p file.paf`, src/seqrush.rs:510-609): replaces sr_ctx_load + sr_ctx_align.
 * Every record of the PAF file whose names are known and that carries a cg:Z: tag is replayed through
 * process_alignment's rules (src/seqrush.rs:1134-1481: bases of M/= ops are compared, runs >= min_match_len
 * are united, query_start/target_start honoured, strand '-' = reverse-complemented query); follow with
 * sr_ctx_unite.  Records are sharded over ranks like the pair list.  sr_unite_paf is the fused form of
 * load + unite + download (raw uf_rush nodes, or canonical labels with p->canonical_labels). */
int sr_ctx_load_paf(sr_ctx *c, const sr_seqset *seqs, const sr_params *p, const char *paf_path);
int sr_unite_paf(const sr_seqset *seqs, const sr_params *p, const char *paf_path, uint64_t *parent_out);
/* graph induction on the device from the context's union-find (node ids in first-
 * encounter order, node bases, path steps, deduplicated edges; src/bidirected_builder.rs:17-289,
 * src/bidirected_ops.rs:813-825) + GFA text formatted on the host.  Byte-identical to sr_build_gfa() on the
 * canonical labels of the same context; *gfa is malloc'ed (sr_free). */
int sr_ctx_build_gfa(sr_ctx *c, const sr_seqset *seqs, char **gfa, uint64_t *n_nodes, uint64_t *n_edges);
/* timing of the last enqueued kernels, measured with hipEvents on the
 * context's stream: which = 0 align (the alignment kernel proper), 1 unite, 2 labels/merge, 3 graph induction,
 * 4 the orientation kernel (when orientation runs as its own kernel before the alignment kernel) */
int sr_ctx_kernel_ms(sr_ctx *c, int which, float *ms);
/* name of the alignment kernel the loaded context launches ("sr_align_blk_kernel",
 * "sr_align_bfs_kernel" or "sr_align_kernel"), NULL before sr_ctx_load */
const char *sr_ctx_align_kernel(const sr_ctx *c);
/* device counters accumulated by the last align: [0] wavefront cells,
 * [1] wavefront steps, [2] base-case segments, [3] breakpoint searches,
 * [4] united bases (after unite), [5] match runs, [6] orientation: wavefront cells of the orientation kernel (or,
 * when orientation runs inside the alignment kernel, its 100 MHz ticks), [7..8] 100 MHz ticks summed
 * over workgroups: breakpoint search, base cases; [9] breakpoint-
 * search passes; [10] ticks of whole pairs; [11..15] ticks inside the breakpoint
 * search: wavefront pass, barrier wait, phase-1 control, breakpoint detection, tail */
int sr_ctx_counters(sr_ctx *c, uint64_t out[16]);
/* the same plus [16] bytes of wavefront rows loaded, [17] stored by the alignment kernel's tiles (every lane access of
 * every tile counted on the device: the kernel's algorithmic HBM bytes, bench.py roofline); [18..31] reserved.
 * The tick counters [6..15] are filled only by the instrumented kernel instance (environment SR_PROFILE_TICKS=1). */
int sr_ctx_counters_ext(sr_ctx *c, uint64_t out[32]);
/* every slot (48 since ABI 2 / round 4): [32..35] tile stamps of the diagnostic build (cycles waiting for a tile's first
 * rows, cycles in its levels, tiles, extension-loop iterations), [36] bytes of wavefront rows that stayed in LDS
 * (LDS-resident histories of small base cases), [37] base cases searched again with the worst-case history region,
 * [40..43] first offending access of the bounds-checked build (pair, level, row offset, extent), rest reserved.
 * Returns the number of slots written (<= cap) or a negative sr_status. */
int sr_ctx_counters_all(sr_ctx *c, uint64_t *out, uint32_t cap);
/* "NAME<TAB>meaning when unset<TAB>what it does" lines of every environment variable the load path reads; the ones that
 * are set are listed under "knobs" in sr_ctx_workspace_report() */
const char *sr_knobs_doc(void);

/* -------- consumer (A9): graph induction + GFA, host C++ -----------------
 * build_bidirected_graph_with_options (bidirected_builder.rs:17-289) +
 * write_gfa (bidirected_ops.rs:880-925) for --no-sort --no-compact, from
 * canonical labels.  Returns malloc'd text in *gfa (free with sr_free). */
int sr_build_gfa(const sr_seqset *seqs, const uint64_t *labels, char **gfa,
                 uint64_t *n_nodes, uint64_t *n_edges);
/* the same with the reference's post-induction step for `--no-sort` without `--no-compact`
 * (src/bidirected_gfa_writer.rs:39-51): compact() merges linear chains of perfect neighbours round after round
 * (src/bidirected_ops.rs:91-490), renumber_nodes_sequentially() (:75-89); compact == 0 is sr_build_gfa */
int sr_build_gfa_opts(const sr_seqset *seqs, const uint64_t *labels, int compact, char **gfa,
                      uint64_t *n_nodes, uint64_t *n_edges);
/* Induction from a RAW uf_rush node array with the reference's root rule (src/bidirected_builder.rs:46-48, 176-182: a
 * node's base is the base at the offset of union_find.find(pos), i.e. of the component's ROOT): the entry point for a
 * host that wants byte equality with a reference run -- node orientation on reverse-complement components included --
 * by replaying the unites in a fixed order (its own uf_rush, or sr_uf_unite_host).  The canonical entry points above
 * relabel every component to its minimum Pos first (schedule-independent). */
int sr_build_gfa_from_nodes(const sr_seqset *seqs, const uint64_t *nodes, int compact, char **gfa,
                            uint64_t *n_nodes, uint64_t *n_edges);
int sr_ctx_build_gfa_opts(sr_ctx *c, const sr_seqset *seqs, int compact, char **gfa, uint64_t *n_nodes, uint64_t *n_edges);
void sr_free(void *p);

const char *sr_last_error(void);
int sr_abi_version(void);
int sr_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
