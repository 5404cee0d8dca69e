// Internal structs shared by the HIP kernels (sr_device.hip) and the host side
// of the C ABI (sr_host.cpp).  Not part of the public interface.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define SR_NULL_OFF (-8192)          // null wavefront offset (fits int16)
#define SR_MAX_SCOPE 127             // max ring depth supported on device
#define SR_STACK_DEPTH 64
#define SR_WG 256                    // threads per workgroup (4 waves of 64)
#define SR_BFS_MAXSEG 4096            // segments per pair in the bfs kernel's ordered list
#define SR_BFS_SEGREC 16             // ints per segment record
#define SR_BFS_MAXACT 32             // segments searched concurrently (2 aligners each)
#define SR_BFS_BTCAP 1024
#define SR_NULL_ROWS 10               // NULL rows of the blocked kernel's ring (a whole block of levels below 0, sr_align_blk.inc kbase)
#define SR_BLK_MAK_SLOTS 80          // ring depth the blocked kernel supports (2 * scope + 2 * block + 2 = 74 for 0,5,8,2,24,1)

enum { SR_C_M = 0, SR_C_I1 = 1, SR_C_I2 = 2, SR_C_D1 = 3, SR_C_D2 = 4 };
// raw WFA2 op codes used in device CIGAR ops: (len << 4) | op
enum { SR_OP_M = 0, SR_OP_X = 1, SR_OP_I = 2 /* consumes text */, SR_OP_D = 3 /* consumes pattern */ };

// device error bits (per launch, OR-ed into *error_flag)
enum {
    SR_DEV_ERR_SCORE_BOUND = 1,   // score loop exceeded its safety bound
    SR_DEV_ERR_BASE_OVERFLOW = 2, // base-case WFA exceeded the history depth
    SR_DEV_ERR_BACKTRACE = 4,     // backtrace found no predecessor
    SR_DEV_ERR_STACK = 8,         // biWFA recursion stack overflow
    SR_DEV_ERR_CIGAR_OVERFLOW = 16,
    SR_DEV_ERR_UF_SPIN = 32,      // union-find retry bound hit
    SR_DEV_ERR_BREAKPOINT = 64,   // breakpoint outside the segment
    SR_DEV_ERR_GRAPH = 128,       // graph induction: strands of a base in different components / hash table full
    SR_DEV_ERR_ADDRESS = 256      // bounds-checked build (-DSR_BOUNDS): a row offset outside the workgroup's extent or an LDS window
                                  // of a live cell outside the staged sequences (the access was not made; counters[40..43])
};

struct SrPen {
    int x, o1, e1, o2, e2;
    int two;      // 1 = gap-affine-2p
    int scope;    // max(x, o1+e1, o2+e2) + 1
};

struct SrAlignArgs {
    // packed sequences: 2 bits/base, 16 bases per word, every sequence copy is
    // framed by one pad word on each side
    const uint32_t *seqwords;
    const uint64_t *word_off_fwd;   // [n] index of the first real word
    const uint64_t *word_off_rc;    // [n] same for the reverse complement
    const uint64_t *word_off_rev;   // [n] reversed copy (2-bit ACGT buffers: == word_off_rc, see sr_align_blk.inc)
    const uint64_t *word_off_cmp;   // [n] complemented copy (2-bit ACGT buffers: == word_off_fwd)
    int symbits;                    // 2, 4 or 8 bits per symbol (selects the kernel build, sr_dev_common.h)
    const uint32_t *order;          // [npairs] dequeue order (cost-sorted), NULL = list order
    const uint32_t *seqlen;         // [n]
    uint32_t max_words;             // words per LDS region (incl. pads)
    // pair list (this rank's shard)
    const uint32_t *pair_q, *pair_t;
    uint32_t npairs;
    uint32_t *queue_head;
    SrPen pen, ori;
    int mem_mode;
    // per-workgroup workspace
    int ring_scope;            // max scope over both penalty sets
    int ring_hot;              // hot I/D rows per component of the level-per-pass kernel's ring (max gap-extend + 2)
    int hist_w, hist_levels;   // a worst-case base case: history columns, levels
    // level-synchronous ("bfs") kernel workspace, per workgroup
    int impl;                  // 1 = sr_align_bfs_kernel (level per pass), 2 = sr_align_blk_kernel (score-blocked wave tiles;
                               // rows: [kdepth][5 components] | NULL row | U row | trash row)
    int kdepth;                // impl 2: ring depth of the M rows (scope + levels per block + 1; lazy I/D rows: 2 scope + 2 blocks + 2)
    int kdepth2;               // impl 2: ring depth of the I / D rows (lazy I/D rows: scope + 2 blocks + 2, else kdepth)
    int kblock;                // impl 2: score levels per block (5: generic instance, 10: exact-penalty instance)
    void *bring;               // rows of brow offsets: M[(ring_scope+1)] | hot I1 I2 D1 D2 [ring_hot] each |
                               //   cold [(ring_scope+1)][4] | NULL row ; every aligner owns a sub-range of each row
    uint64_t bring_wg_stride;
    int brow;
    void *bhist;               // level-per-pass kernel: [hist_levels][5][bbase_jobs * hist_w] + NULL row; blocked kernel: hist_cap cells
                               // in which every base case of a batch gets [its levels][5][its width], + a NULL row
    uint64_t bhist_wg_stride;
    uint64_t hist_cap;         // blocked kernel: cells of a workgroup's history (>= hist_levels * 5 * hist_w: one worst-case base case)
    uint32_t hist_stride;      // level-per-pass kernel: cells per history row (bbase_jobs * hist_w)
    uint32_t hist_nul_w;       // blocked kernel: cells of the NULL row (a worst-case job's width + read slack)
    int bbase_jobs;
    int *bseg;                 // 2 segment lists of SR_BFS_MAXSEG records x SR_BFS_SEGREC ints
    uint32_t *bbt;             // [bbase_jobs][SR_BFS_BTCAP] reversed run-length ops of finished base cases
    uint32_t *bcl;             // impl 2: breakpoint candidate list, bcl_wg_stride entries per workgroup
    uint64_t bcl_wg_stride;
    // orientation as its own kernel (sr_orient.hip): one pair per wave
    void *oring;               // per 64-thread workgroup: [ori.scope+1 levels][M,I1,D1] rows of orow offsets + NULL row
    uint64_t oring_wg_stride;
    int orow;
    uint32_t *oqueue;          // pair queue of the orientation kernel
    const uint32_t *kbits;     // sr_orient_blk_kernel, 2-bit buffers: per sequence a 2^16-bit set of the 8-mers of its forward copy
                               //   (q-gram bound of the reverse-complement orientation's score), NULL = none
    int pre_oriented;          // 1: is_reverse / ori_fwd / ori_rev are inputs of the alignment kernel
    int lazy_id;               // impl 2: searches in phase 1 do not store the I/D rows only breakpoint detection reads
                               //   (kdepth >= 2 * scope + 2 * block + 2 so that they can be recomputed)
    int profile_ticks;         // impl 2: launch the instrumented instance (SR_PROFILE_TICKS=1)
    int ori_levels;            // impl 2: in-kernel orientation level by level even for the default penalties (SR_ORIENT_LEVELS=1)
    int ring_u16;              // impl 2, 32-bit searches: the ring's cells are uint16 = offset + 8192 (longest sequence < 57 k)
    int *bmak;                 // impl 2: per workgroup [32 aligners][32 ring levels] max M antidiagonal (breakpoint pruning)
    int test_base_levels;      // impl 2, tests only (SR_TEST_BASE_LEVELS=n): cap on the levels a base case is given at first, so that
                               //   jobs outgrow their region and take the re-queue path
    // impl 2, sr_ctx_run: the workgroup unites the bases of a pair's match runs right after it emitted the CIGAR
    // (sr_uf_dev.h uf_unite_cigar; no sr_unite_kernel launch for the batch)
    int fuse_unite;
    unsigned long long *uf_nodes;   // uf_rush node array
    const uint64_t *seq_goff;       // [n] global offsets (concatenated coordinates)
    const int32_t *max_score;       // [npairs] divergence filter bound or INT_MAX
    uint64_t min_match_len;
    // outputs
    uint8_t *is_reverse;       // [npairs]
    int32_t *score;            // [npairs]
    int32_t *ori_fwd, *ori_rev;// [npairs] orientation scores (rev = INT_MAX if not better)
    uint32_t *cigar_ops;
    const uint64_t *cigar_base;// [npairs+1]
    uint32_t *cigar_cnt;       // [npairs]
    unsigned long long *counters; // [SR_NCOUNTERS]
    int *error_flag;
};
#define SR_NCOUNTERS 48

struct SrUniteArgs {
    const uint32_t *pair_q, *pair_t;
    uint32_t npairs;
    const uint32_t *seqlen;
    const uint64_t *seq_goff;       // [n] global offsets (concatenated coordinates)
    const uint32_t *q_start, *t_start; // [npairs] first aligned position (alignment orientation) or NULL = 0 (PAF input)
    const uint8_t *is_reverse;
    const int32_t *score;
    const int32_t *max_score;       // [npairs] divergence filter bound or INT_MAX
    const uint32_t *cigar_ops;
    const uint64_t *cigar_base;
    const uint32_t *cigar_cnt;
    uint64_t min_match_len;
    unsigned long long *nodes;      // uf_rush node array
    uint64_t uf_size;
    unsigned long long *counters;
    int *error_flag;
};

#ifdef __cplusplus
extern "C" {
#endif
// launchers implemented in sr_device.hip (hipStream_t passed as void*)
int srk_align(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream);   // by a->impl, a->symbits
int srk_unite(const SrUniteArgs *a, int nwg, void *stream);
int srk_uf_init(unsigned long long *nodes, uint64_t total_len, uint64_t uf_size, void *stream);
int srk_labels(unsigned long long *nodes, uint64_t uf_size, unsigned long long *minarr,
               unsigned long long *labels, int *error_flag, void *stream);
int srk_merge(unsigned long long *nodes, uint64_t uf_size, const unsigned long long *labels,
              uint32_t count, int *error_flag, void *stream);
int srk_align_max_lds(void);
int srk_orient(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, void *stream);
int srk_kmer_bits(const SrAlignArgs *a, uint32_t nseq, uint32_t *kbits, void *stream);      // build SrAlignArgs::kbits (2-bit buffers)
// sr_order.hip: dequeue order of a batch by predicted cost (orientation score x length), sorted on the device
size_t srk_order_temp_bytes(uint32_t n);
int srk_order(const SrAlignArgs *a, uint64_t *keys_in, uint64_t *keys_out, uint32_t *vals_in, void *temp, size_t temp_bytes,
              uint32_t *order_out, void *stream);
int srk_graph_induce(const unsigned long long *labels, const uint8_t *bases, const uint8_t *islast,
                     uint64_t N, uint64_t uf_size, unsigned long long *first, uint32_t *flag, uint32_t *nid,
                     uint32_t *steps, uint8_t *node_base, unsigned long long *hkeys, uint32_t *hvals,
                     uint64_t hcap, uint32_t *eslot, unsigned long long *edges, uint32_t *tile_sum,
                     uint32_t *counts, int *error_flag, void *stream);
const char *srk_source_digest(void);                                 // digest of the sources this library was built from (Makefile, scripts/src_digest.py)
const char *srk_align_blk_build_tag(void);                         // name of the blocked kernel's build (scripts/build_variant.sh)
int srk_align_blk_max_levels(void);                                // deepest block of the build (static LDS tables)
int srk_align_blk_supports(const SrPen *pen, const SrPen *ori);   // levels per block, 0 = no blocked instance
int srk_labels32(unsigned long long *nodes, uint64_t uf_size, unsigned long long *minarr, uint32_t *labels, int *error_flag,
                 void *stream);
int srk_merge32(unsigned long long *nodes, uint64_t uf_size, const uint32_t *labels, uint32_t count, int *error_flag, void *stream);
// sr_sketch.hip: k-mer bottom-s sketches, all-pairs similarity, k-nearest / k-farthest selection
int srk_sketch(const uint8_t *bases, const uint64_t *goff, const uint32_t *len, uint32_t n, int k, int s_max,
               unsigned long long *scratch, uint64_t stride, const uint32_t *npad, unsigned long long *sketch, uint32_t *sk_n,
               void *stream);
int srk_jaccard(const unsigned long long *sketch, const uint32_t *sk_n, uint32_t n, int s_max, uint32_t *shared, uint32_t *denom,
                void *stream);
int srk_knn_select(const uint32_t *shared, const uint32_t *denom, uint32_t n, int kn, int kf, uint8_t *sel, void *stream);
#ifdef __cplusplus
}
#endif
