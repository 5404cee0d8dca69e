/*
 * sr_oracle.h -- CPU ORACLE for the seqrush hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain-C restatement of the reference's algorithm for the
 * path  all-vs-all alignment -> CIGAR match-run extraction -> bidirected
 * union-find unite() -> (consumer) graph induction + GFA.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it;
 * the product library (seqrush_amd/csrc) never does.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  Where the algorithm lives in a dependency whose
 * source is absent from the reference tree (allwave 0.1.0, lib_wfa2 @819b82c
 * -> WFA2-lib) the published algorithm is restated (Marco-Sola et al. 2021
 * "Fast gap-affine pairwise alignment using the wavefront algorithm"; 2023
 * "Optimal gap-affine alignment in O(s) space") and anchored on the
 * reference's own call sites and tests (see DESIGN.md, section "Oracle").
 *
 * PARITY STATUS: the in-tree parts (Pos, uf_rush, BidirectedUnionFind,
 * process_alignment, graph induction, GFA writer, FASTA/score parsers) are
 * pinned by the reference's own unit-test known answers (tests/golden/).  The
 * WFA part is pinned on optimal score (independent Gotoh DP, WFA v1 C code
 * shipped in the reference's cargo cache) and on the reference's WFA2
 * boundary tests; for co-optimal tie-breaking, biWFA breakpoints and the
 * orientation rule: PARITY UNPINNED (no WFA2-lib source, no golden CIGAR of a
 * divergent pair exists in the reference).
 */
#ifndef SR_ORACLE_H
#define SR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- Pos codec: src/pos.rs:6-87 ---------------- */
typedef uint64_t sro_pos;
sro_pos sro_make_pos(uint64_t offset, int is_reverse);   /* pos.rs:10-12 */
int sro_is_rev(sro_pos p);                               /* pos.rs:16-18 */
uint64_t sro_offset(sro_pos p);                          /* pos.rs:22-24 */
sro_pos sro_incr_pos(sro_pos p);                         /* pos.rs:28-41 */
sro_pos sro_decr_pos(sro_pos p);                         /* pos.rs:45-58 */
sro_pos sro_flip_orientation(sro_pos p);                 /* pos.rs:62-64 */
char sro_orientation_char(sro_pos p);                    /* pos.rs:68-74 */
uint8_t sro_rc_base(uint8_t b);                          /* pos.rs:78-87 */

/* ---------------- uf_rush 0.2.1 (vendored crate src/lib.rs) ------------ */
typedef struct sro_uf sro_uf;
sro_uf *sro_uf_new(uint64_t size);                       /* lib.rs:36-42 */
void sro_uf_free(sro_uf *u);
uint64_t sro_uf_size(const sro_uf *u);                   /* lib.rs:48-50 */
uint64_t sro_uf_find(sro_uf *u, uint64_t x);             /* lib.rs:112-133 */
int sro_uf_unite(sro_uf *u, uint64_t x, uint64_t y);     /* lib.rs:159-208 */
int sro_uf_same(sro_uf *u, uint64_t x, uint64_t y);      /* lib.rs:72-84 */
uint64_t *sro_uf_nodes(sro_uf *u);   /* raw packed parent|rank<<58 array */

/* ------------- BidirectedUnionFind: src/bidirected_union_find.rs ------- */
sro_uf *sro_buf_new(uint64_t max_offset);                /* :16-24 */
sro_pos sro_buf_find(sro_uf *u, sro_pos p);              /* :27-31 */
void sro_buf_unite(sro_uf *u, sro_pos a, sro_pos b);     /* :35-43 */
int sro_buf_same(sro_uf *u, sro_pos a, sro_pos b);       /* :46-54 */
void sro_buf_unite_matching_region(sro_uf *u, uint64_t seq1_offset,
    uint64_t seq2_offset, uint64_t seq1_local_start, uint64_t seq2_local_start,
    uint64_t match_length, int seq1_is_rc, uint64_t seq1_len);   /* :60-98 */
void sro_buf_unite_matching_region_seq2_rc(sro_uf *u, uint64_t seq1_offset,
    uint64_t seq2_offset, uint64_t seq1_local_start, uint64_t seq2_local_start,
    uint64_t match_length, int seq2_is_rc, uint64_t seq2_len);   /* :102-129 */

/* ---------------- penalties / score strings ---------------- */
typedef struct {
    int32_t match;      /* must be 0 (reference default, seqrush.rs:45) */
    int32_t mismatch;   /* x  */
    int32_t gap_open1;  /* o1 */
    int32_t gap_ext1;   /* e1 */
    int32_t gap_open2;  /* o2, <0 = single-piece affine */
    int32_t gap_ext2;   /* e2 */
} sro_penalties;
/* AlignmentScores::parse seqrush.rs:165-217; returns 0 or <0 */
int sro_parse_scores(const char *s, sro_penalties *out);
/* parse_orientation seqrush.rs:219-250 */
int sro_parse_orientation_scores(const char *s, sro_penalties *out);
/* max_score_for_divergence seqrush.rs:253-269 */
int32_t sro_max_score_for_divergence(const sro_penalties *p, uint64_t seq_len,
                                     double max_divergence);

/* sparsification grammar, seqrush.rs:356-431 */
enum { SRO_SPARSE_NONE = 0, SRO_SPARSE_AUTO = 1, SRO_SPARSE_RANDOM = 2,
       SRO_SPARSE_CONNECTIVITY = 3, SRO_SPARSE_TREE = 4 };
typedef struct {
    int kind; double factor; uint64_t k_nearest, k_farthest; double rand_frac;
    uint64_t kmer_size;
} sro_sparsification;
int sro_parse_sparsification(const char *s, sro_sparsification *out);

/* ---------------- WFA (restated WFA2-lib; see wfa.c header) ------------- */
enum { SRO_M = 0, SRO_I1 = 1, SRO_I2 = 2, SRO_D1 = 3, SRO_D2 = 4 };
enum { SRO_MEM_HIGH = 0, SRO_MEM_ULTRALOW = 3 };   /* lib_wfa2 MemoryMode */
#define SRO_BIALIGN_FALLBACK_MIN_SCORE 250
#define SRO_BIALIGN_FALLBACK_MIN_LENGTH 100

/* Raw WFA2-alphabet CIGAR, one byte per column: 'M' match, 'X' mismatch,
 * 'I' consumes text(target), 'D' consumes pattern(query)
 * (tests/test_cigar_validity.rs:90-99).  Caller frees *cigar with free(). */
int sro_wfa_align(const uint8_t *pattern, int plen, const uint8_t *text,
                  int tlen, const sro_penalties *pen, int memory_mode,
                  uint8_t **cigar, int *cigar_len, int *score);
/* Score only; max_score<0 = unbounded, else returns INT32_MAX in *score when
 * the optimal score exceeds max_score. */
int sro_wfa_score(const uint8_t *pattern, int plen, const uint8_t *text,
                  int tlen, const sro_penalties *pen, int max_score,
                  int *score);
/* number of (score,diagonal) wavefront cells computed by the last
 * sro_wfa_align on this thread (all components of one cell count once) */
uint64_t sro_wfa_last_cells(void);
/* Independent O(n*m) Gotoh two-piece affine DP (score oracle for the WFA) */
int sro_gotoh_score(const uint8_t *pattern, int plen, const uint8_t *text,
                    int tlen, const sro_penalties *pen);
/* Score of a raw CIGAR under pen; -1 if it does not spell pattern/text */
int sro_cigar_score(const uint8_t *cigar, int n, const uint8_t *pattern,
                    int plen, const uint8_t *text, int tlen,
                    const sro_penalties *pen);

/* cigar_bytes_to_string: src/wfa.rs:9-38.  Returns malloc'd C string. */
char *sro_cigar_bytes_to_string(const uint8_t *cigar_bytes, int n);

/* ---------------- sequences / SeqRush ---------------- */
typedef struct {
    char *id;
    uint8_t *data;
    uint64_t len;
    uint64_t offset;
} sro_sequence;
typedef struct {
    sro_sequence *seqs;
    uint64_t n;
    uint64_t total_length;
    sro_uf *uf;
} sro_seqrush;

/* load_sequences seqrush.rs:1801-1837 (from memory buffer / file) */
int sro_load_fasta_mem(const char *text, size_t n, sro_sequence **out,
                       uint64_t *count);
int sro_load_fasta(const char *path, sro_sequence **out, uint64_t *count);
void sro_free_sequences(sro_sequence *s, uint64_t n);
/* Sequence::reverse_complement seqrush.rs:281-295 */
void sro_reverse_complement(const uint8_t *in, uint64_t n, uint8_t *out);

/* SeqRush::new seqrush.rs:308-336 (takes ownership of seqs). NULL + errbuf on
 * empty sequence ("Empty sequences are not allowed", :310-317). */
sro_seqrush *sro_seqrush_new(sro_sequence *seqs, uint64_t n, char *err,
                             size_t errlen);
void sro_seqrush_free(sro_seqrush *s);
uint64_t sro_count_components(sro_seqrush *s);           /* :341-353 */

/* process_alignment seqrush.rs:1134-1481. Returns #united bases or <0 when
 * validate_match (:1179-1207) would have panicked. */
int64_t sro_process_alignment(sro_seqrush *s, const char *cigar, uint64_t q_idx,
    uint64_t t_idx, uint64_t min_match_len, int query_is_rc,
    uint64_t query_start, uint64_t query_end, uint64_t target_start,
    uint64_t target_end);

/* ---------------- pair driver (restated allwave behaviour) ------------- */
typedef struct {
    uint32_t query_idx, target_idx;
    int is_reverse;
    int score;
    uint8_t *cigar_bytes;   /* raw WFA2 alphabet */
    int cigar_len;
    uint64_t query_start, query_end, target_start, target_end;
} sro_alignment;
typedef struct {
    sro_penalties pen;          /* -S, default 0,5,8,2,24,1 */
    sro_penalties ori;          /* --orientation-scores, default 0,1,1,1 */
    uint64_t min_match_len;     /* -k */
    double max_divergence;      /* -d, <0 = none */
    int exclude_self;           /* reference passes false (seqrush.rs:731) */
    int memory_mode;            /* SRO_MEM_ULTRALOW in the reference (wfa.rs:57) */
    int threads;                /* -t, default 4 (seqrush.rs:37) */
} sro_params;
void sro_default_params(sro_params *p);
/* orientation + full alignment of one ordered pair (q,t) */
int sro_align_pair(const sro_seqrush *s, const sro_params *p, uint32_t q,
                   uint32_t t, sro_alignment *out);
void sro_alignment_free(sro_alignment *a);
/* align_and_unite_with_allwave seqrush.rs:611-757 over pairs
 * [pair_begin,pair_end) of the row-major n*n ordered pair list. Returns the
 * number of alignments processed; *cells gets DP-equivalent cells. */
int64_t sro_align_and_unite(sro_seqrush *s, const sro_params *p,
                            uint64_t pair_begin, uint64_t pair_end,
                            uint64_t *dp_cells);

/* the same over an explicit ordered pair list */
int64_t sro_align_and_unite_list(sro_seqrush *s, const sro_params *p, const uint32_t *pq, const uint32_t *pt,
                                 uint64_t count, uint64_t *dp_cells);
/* the same, keeping per-pair score / strand / number of CIGAR runs / run digest (full-size comparisons in tests/);
 * do_unite = 0: align only */
int64_t sro_align_and_unite_list_collect(sro_seqrush *s, const sro_params *p, const uint32_t *pq, const uint32_t *pt,
                                         uint64_t count, int do_unite, int32_t *score, uint8_t *is_reverse,
                                         uint32_t *cigar_runs, uint64_t *cigar_digest);
uint64_t sro_cigar_run_digest(const uint8_t *raw_cigar, uint64_t n);
/* sparsified ordered pair list (own definition, see seqrush.c; PARITY UNPINNED: allwave's rules are absent).
 * Arrays are malloc'd (free()). */
int sro_sparsified_pairs(const sro_seqrush *s, const sro_sparsification *sp, uint64_t seed, int exclude_self,
                         uint32_t **pq_out, uint32_t **pt_out, uint64_t *count);

/* ---------------- graph induction + GFA (consumer, A9) ---------------- */
/* build_bidirected_graph_with_options bidirected_builder.rs:17-289 +
 * write_gfa bidirected_ops.rs:880-925, --no-sort --no-compact.
 * canonical!=0: every UF component is first relabelled to its minimum Pos
 * (project decision, DESIGN.md: the reference's root identity depends on the
 * rayon schedule).  L lines are emitted in first-insertion order.
 * Returns malloc'd GFA text. */
char *sro_build_gfa(sro_seqrush *s, int canonical, int faithful_scan,
                    uint64_t *n_nodes, uint64_t *n_edges);
/* compact() + renumber_nodes_sequentially() (src/bidirected_ops.rs:75-490) on a --no-compact GFA text as
 * sro_build_gfa writes it: the graph the reference writes for --no-sort without --no-compact
 * (src/bidirected_gfa_writer.rs:39-51).  Literal restatement, quadratic: test-sized graphs only.  malloc'd text. */
/* Handle codec (src/bidirected_graph.rs:9-64) */
uint64_t sro_handle_new(uint64_t node_id, int is_reverse);
uint64_t sro_handle_node_id(uint64_t h);
int sro_handle_is_reverse(uint64_t h);
char sro_handle_orientation_char(uint64_t h);
uint64_t sro_handle_flip(uint64_t h);
char *sro_compact_gfa(const char *gfa, uint64_t *n_nodes, uint64_t *n_edges);
/* parse + write_gfa (ops:880-925) without compaction; spell a path (bidirected_graph.rs:113-154) */
char *sro_rewrite_gfa(const char *gfa, uint64_t *n_nodes, uint64_t *n_edges);
char *sro_gfa_path_sequence(const char *gfa, uint64_t index);
/* canonical min-Pos label per element of the UF (len = uf size) */
void sro_canonical_labels(sro_seqrush *s, uint64_t *labels);

#ifdef __cplusplus
}
#endif
#endif
