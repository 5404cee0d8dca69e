/* ORACLE (test infrastructure) -- SeqRush host logic restated in C:
 * score/sparsification parsers, FASTA loader, SeqRush::new,
 * process_alignment, the allwave pair driver (restated behaviour, source
 * absent) and graph induction + GFA writer.  See sr_oracle.h for the rules. */
#include "sr_oracle.h"
#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* score strings: AlignmentScores::parse  seqrush.rs:165-217           */
/* ------------------------------------------------------------------ */
static int parse_i32(const char *s, size_t n, int32_t *out) {
    /* Rust str::parse::<i32>: optional sign, digits only, no whitespace */
    if (n == 0 || n > 15) return -1;
    char buf[16];
    memcpy(buf, s, n); buf[n] = 0;
    size_t i = 0;
    if (buf[0] == '+' || buf[0] == '-') i = 1;
    if (i == n) return -1;
    for (size_t j = i; j < n; j++) if (!isdigit((unsigned char)buf[j])) return -1;
    long long v = atoll(buf);
    if (v > INT32_MAX || v < INT32_MIN) return -1;
    *out = (int32_t)v;
    return 0;
}

static int split_commas(const char *s, const char **parts, size_t *lens, int maxp) {
    int n = 0;
    const char *start = s;
    for (const char *c = s;; c++) {
        if (*c == ',' || *c == 0) {
            if (n < maxp) { parts[n] = start; lens[n] = (size_t)(c - start); }
            n++;
            if (*c == 0) break;
            start = c + 1;
        }
    }
    return n;
}

int sro_parse_scores(const char *s, sro_penalties *out) {
    const char *parts[8]; size_t lens[8];
    int n = split_commas(s, parts, lens, 8);
    if (n < 4) return -1;              /* :168-173 */
    if (n > 6) return -2;              /* :175-177 */
    int32_t v[6] = {0, 0, 0, 0, -1, -1};
    for (int i = 0; i < 4; i++)
        if (parse_i32(parts[i], lens[i], &v[i])) return -3 - i;
    if (n >= 6) {                      /* :192-207 (5 values => no 2nd piece) */
        if (parse_i32(parts[4], lens[4], &v[4])) return -7;
        if (parse_i32(parts[5], lens[5], &v[5])) return -8;
    }
    out->match = v[0]; out->mismatch = v[1]; out->gap_open1 = v[2];
    out->gap_ext1 = v[3]; out->gap_open2 = v[4]; out->gap_ext2 = v[5];
    return 0;
}

int sro_parse_orientation_scores(const char *s, sro_penalties *out) { /* :219-250 */
    const char *parts[8]; size_t lens[8];
    int n = split_commas(s, parts, lens, 8);
    if (n != 4) return -1;
    int32_t v[4];
    for (int i = 0; i < 4; i++)
        if (parse_i32(parts[i], lens[i], &v[i])) return -3 - i;
    out->match = v[0]; out->mismatch = v[1]; out->gap_open1 = v[2];
    out->gap_ext1 = v[3]; out->gap_open2 = -1; out->gap_ext2 = -1;
    return 0;
}

int32_t sro_max_score_for_divergence(const sro_penalties *p, uint64_t seq_len,
                                     double max_divergence) {      /* :253-269 */
    int32_t max_mismatches = (int32_t)ceil((double)seq_len * max_divergence);
    int32_t max_gaps = (int32_t)ceil((double)seq_len * max_divergence * 0.5);
    int32_t mismatch_score = max_mismatches * p->mismatch;
    int32_t gap_score = max_gaps > 0 ? p->gap_open1 + (max_gaps - 1) * p->gap_ext1 : 0;
    int32_t threshold = mismatch_score + gap_score;
    int32_t floor_ = p->mismatch * 2;
    return threshold > floor_ ? threshold : floor_;
}

/* parse_sparsification seqrush.rs:356-431 */
static int parse_f64(const char *s, double *out) {
    if (!*s) return -1;
    for (const char *q = s; *q; q++) if (*q == 'x' || *q == 'X') return -1;   /* strtod takes hex floats, Rust's f64 parse does not */
    char *end;
    double v = strtod(s, &end);
    if (*end != 0 || isspace((unsigned char)s[0])) return -1;
    *out = v;
    return 0;
}
static int parse_usize(const char *s, size_t n, uint64_t *out) {
    if (n == 0) return -1;
    uint64_t v = 0;
    size_t i = 0;
    if (s[0] == '+') { i = 1; if (n == 1) return -1; }
    for (; i < n; i++) {
        if (!isdigit((unsigned char)s[i])) return -1;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (UINT64_MAX - d) / 10) return -1;         /* usize overflow is an Err in Rust */
        v = v * 10 + d;
    }
    *out = v;
    return 0;
}

int sro_parse_sparsification(const char *s, sro_sparsification *out) {
    memset(out, 0, sizeof(*out));
    out->kmer_size = 16;
    if (!strcmp(s, "none") || !strcmp(s, "1.0")) { out->kind = SRO_SPARSE_NONE; return 0; }
    if (!strcmp(s, "auto")) { out->kind = SRO_SPARSE_AUTO; return 0; }
    if (!strncmp(s, "random:", 7)) {
        double f;
        if (parse_f64(s + 7, &f)) return -1;
        if (f > 0.0 && f <= 1.0) { out->kind = SRO_SPARSE_RANDOM; out->factor = f; return 0; }
        return -2;
    }
    if (!strncmp(s, "connectivity:", 13)) {
        double f;
        if (parse_f64(s + 13, &f)) return -1;
        if (f > 0.0 && f <= 1.0) { out->kind = SRO_SPARSE_CONNECTIVITY; out->factor = f; return 0; }
        return -2;
    }
    if (!strncmp(s, "tree:", 5)) {
        const char *parts[8]; size_t lens[8];
        int n = split_commas(s + 5, parts, lens, 8);
        if (n < 1 || n > 4) return -3;
        if (parse_usize(parts[0], lens[0], &out->k_nearest)) return -4;
        out->k_farthest = 0; out->rand_frac = 0.0; out->kmer_size = 16;
        if (n >= 2 && parse_usize(parts[1], lens[1], &out->k_farthest)) return -5;
        if (n >= 3) {
            char buf[64];
            if (lens[2] >= sizeof(buf)) return -6;
            memcpy(buf, parts[2], lens[2]); buf[lens[2]] = 0;
            if (parse_f64(buf, &out->rand_frac)) return -6;
            if (out->rand_frac < 0.0 || out->rand_frac > 1.0) return -6;
        }
        if (n >= 4) {
            if (parse_usize(parts[3], lens[3], &out->kmer_size)) return -7;
            if (out->kmer_size == 0) return -7;
        }
        out->kind = SRO_SPARSE_TREE;
        return 0;
    }
    double f;   /* backward compatibility: plain float = random factor (:423-429) */
    if (!parse_f64(s, &f) && f > 0.0 && f <= 1.0) {
        out->kind = SRO_SPARSE_RANDOM; out->factor = f; return 0;
    }
    return -9;
}

/* ------------------------------------------------------------------ */
/* FASTA: load_sequences seqrush.rs:1801-1837                           */
/* ------------------------------------------------------------------ */
static int is_rust_ws(unsigned char c) {
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c;
}

int sro_load_fasta_mem(const char *text, size_t n, sro_sequence **out,
                       uint64_t *count) {
    sro_sequence *seqs = NULL;
    uint64_t ns = 0, cap = 0;
    char *cur_id = NULL;
    uint8_t *cur = NULL;
    size_t cur_len = 0, cur_cap = 0;
    uint64_t offset = 0;
    size_t i = 0;
    while (i < n) {
        size_t j = i;
        while (j < n && text[j] != '\n') j++;
        size_t ls = i, le = j;               /* BufRead::lines strips \n and \r\n */
        if (le > ls && text[le - 1] == '\r') le--;
        if (le > ls && text[ls] == '>') {
            if (cur_id && cur_id[0]) {       /* :1812-1820 */
                if (ns == cap) { cap = cap ? cap * 2 : 16; seqs = (sro_sequence *)realloc(seqs, cap * sizeof(*seqs)); }
                seqs[ns].id = cur_id; seqs[ns].data = cur; seqs[ns].len = cur_len;
                seqs[ns].offset = offset; ns++;
                offset += cur_len;
                cur = NULL; cur_len = 0; cur_cap = 0; cur_id = NULL;
            }
            /* NB: when the previous id was empty, its data is NOT cleared
             * (current_data.clear() sits inside the if, :1819) */
            free(cur_id);
            size_t a = ls + 1;
            while (a < le && is_rust_ws((unsigned char)text[a])) a++;
            size_t b = a;
            while (b < le && !is_rust_ws((unsigned char)text[b])) b++;
            cur_id = (char *)malloc(b - a + 1);
            memcpy(cur_id, text + a, b - a); cur_id[b - a] = 0;   /* :1822 */
        } else {
            size_t a = ls, b = le;           /* line.trim() :1824 */
            while (a < b && is_rust_ws((unsigned char)text[a])) a++;
            while (b > a && is_rust_ws((unsigned char)text[b - 1])) b--;
            if (cur_len + (b - a) > cur_cap) {
                cur_cap = (cur_len + (b - a)) * 2 + 64;
                cur = (uint8_t *)realloc(cur, cur_cap);
            }
            memcpy(cur + cur_len, text + a, b - a);
            cur_len += b - a;
        }
        i = j + 1;
    }
    if (cur_id && cur_id[0]) {               /* :1828-1834 */
        if (ns == cap) { cap = cap ? cap * 2 : 16; seqs = (sro_sequence *)realloc(seqs, cap * sizeof(*seqs)); }
        seqs[ns].id = cur_id; seqs[ns].data = cur ? cur : (uint8_t *)malloc(1);
        seqs[ns].len = cur_len; seqs[ns].offset = offset; ns++;
    } else { free(cur_id); free(cur); }
    for (uint64_t q = 0; q < ns; q++) if (!seqs[q].data) seqs[q].data = (uint8_t *)malloc(1);
    *out = seqs; *count = ns;
    return 0;
}

int sro_load_fasta(const char *path, sro_sequence **out, uint64_t *count) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    size_t got = fread(buf, 1, (size_t)n, f);
    fclose(f);
    int r = sro_load_fasta_mem(buf, got, out, count);
    free(buf);
    return r;
}

void sro_free_sequences(sro_sequence *s, uint64_t n) {
    if (!s) return;
    for (uint64_t i = 0; i < n; i++) { free(s[i].id); free(s[i].data); }
    free(s);
}

void sro_reverse_complement(const uint8_t *in, uint64_t n, uint8_t *out) { /* :281-295 */
    for (uint64_t i = 0; i < n; i++) {
        uint8_t b = in[n - 1 - i];
        switch (b) {
        case 'A': case 'a': b = 'T'; break;
        case 'T': case 't': b = 'A'; break;
        case 'C': case 'c': b = 'G'; break;
        case 'G': case 'g': b = 'C'; break;
        default: break;
        }
        out[i] = b;
    }
}

/* ------------------------------------------------------------------ */
/* SeqRush::new seqrush.rs:308-336                                      */
/* ------------------------------------------------------------------ */
sro_seqrush *sro_seqrush_new(sro_sequence *seqs, uint64_t n, char *err,
                             size_t errlen) {
    for (uint64_t i = 0; i < n; i++) {
        if (seqs[i].len == 0) {               /* :310-317 */
            if (err) snprintf(err, errlen,
                "Empty sequences are not allowed: sequence '%s' has length 0", seqs[i].id);
            return NULL;
        }
    }
    sro_seqrush *s = (sro_seqrush *)calloc(1, sizeof(*s));
    s->seqs = seqs; s->n = n;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; i++) total += seqs[i].len;
    s->total_length = total;
    s->uf = sro_buf_new(total);
    for (uint64_t i = 0; i < total; i++)      /* :324-328 */
        sro_buf_unite(s->uf, sro_make_pos(i, 0), sro_make_pos(i, 1));
    return s;
}

void sro_seqrush_free(sro_seqrush *s) {
    if (!s) return;
    sro_free_sequences(s->seqs, s->n);
    sro_uf_free(s->uf);
    free(s);
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

uint64_t sro_count_components(sro_seqrush *s) {                 /* :341-353 */
    uint64_t n = s->total_length;
    uint64_t *reps = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    for (uint64_t i = 0; i < n; i++) reps[i] = sro_buf_find(s->uf, sro_make_pos(i, 0));
    qsort(reps, n, sizeof(uint64_t), cmp_u64);
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; i++) if (i == 0 || reps[i] != reps[i - 1]) c++;
    free(reps);
    return c;
}

/* ------------------------------------------------------------------ */
/* process_alignment seqrush.rs:1134-1481                               */
/* ------------------------------------------------------------------ */
typedef struct {
    sro_seqrush *s; const sro_sequence *seq1, *seq2; int query_is_rc;
    int64_t united; int failed;
} pa_ctx;

static inline uint8_t pa_query_base(const pa_ctx *c, uint64_t local_pos) { /* :1162-1176 */
    if (c->query_is_rc) {
        uint8_t base = c->seq1->data[c->seq1->len - 1 - local_pos];
        switch (base) {
        case 'A': case 'a': return 'T';
        case 'T': case 't': return 'A';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return base;
        }
    }
    return c->seq1->data[local_pos];
}

static void pa_flush(pa_ctx *c, uint64_t start1, uint64_t start2, uint64_t len) {
    /* validate_match :1179-1207 (panic -> failed flag) */
    for (uint64_t i = 0; i < len; i++) {
        if (pa_query_base(c, start1 + i) != c->seq2->data[start2 + i]) { c->failed = 1; return; }
    }
    sro_buf_unite_matching_region(c->s->uf, c->seq1->offset, c->seq2->offset,
        start1, start2, len, c->query_is_rc,
        c->query_is_rc ? c->seq1->len : 0);
    c->united += (int64_t)len;
}

int64_t sro_process_alignment(sro_seqrush *s, const char *cigar, uint64_t q_idx,
    uint64_t t_idx, uint64_t min_match_len, int query_is_rc,
    uint64_t query_start, uint64_t query_end, uint64_t target_start,
    uint64_t target_end) {
    (void)query_end; (void)target_end;   /* unused by the reference too */
    pa_ctx c = { s, &s->seqs[q_idx], &s->seqs[t_idx], query_is_rc, 0, 0 };
    uint64_t pos1 = query_start, pos2 = target_start, count = 0;   /* :1210-1212 */
    int in_match_run = 0;
    uint64_t run_start1 = 0, run_start2 = 0, run_len = 0;
    const uint64_t len1 = c.seq1->len, len2 = c.seq2->len;
    for (const char *p = cigar; *p; p++) {
        char ch = *p;
        if (ch >= '0' && ch <= '9') { count = count * 10 + (uint64_t)(ch - '0'); continue; }
        if (count == 0) count = 1;                                  /* :1251-1253 */
        if (ch == 'M' || ch == '=') {
            for (uint64_t k = 0; k < count; k++) {
                if (pos1 + k < len1 && pos2 + k < len2) {           /* :1268 */
                    uint8_t b1 = pa_query_base(&c, pos1 + k);
                    uint8_t b2 = c.seq2->data[pos2 + k];
                    if (b1 == b2) {
                        if (!in_match_run) {
                            in_match_run = 1; run_start1 = pos1 + k;
                            run_start2 = pos2 + k; run_len = 1;
                        } else run_len++;
                    } else {
                        if (in_match_run && run_len >= min_match_len)   /* :1311 */
                            pa_flush(&c, run_start1, run_start2, run_len);
                        in_match_run = 0; run_len = 0;
                    }
                }
            }
            pos1 += count; pos2 += count;
        } else {
            if (in_match_run && run_len >= min_match_len)           /* :1365 */
                pa_flush(&c, run_start1, run_start2, run_len);
            in_match_run = 0; run_len = 0;
            if (ch == 'X') { pos1 += count; pos2 += count; }
            else if (ch == 'I') pos1 += count;                      /* :1420-1426 */
            else if (ch == 'D') pos2 += count;                      /* :1427-1433 */
        }
        count = 0;
        if (c.failed) return -1;
    }
    if (in_match_run && run_len >= min_match_len)                   /* :1447 */
        pa_flush(&c, run_start1, run_start2, run_len);
    if (c.failed) return -1;
    return c.united;
}

/* ------------------------------------------------------------------ */
/* pair driver: restated allwave behaviour (source absent) */
/* ------------------------------------------------------------------ */
void sro_default_params(sro_params *p) {
    memset(p, 0, sizeof(*p));
    sro_parse_scores("0,5,8,2,24,1", &p->pen);              /* seqrush.rs:45 */
    sro_parse_orientation_scores("0,1,1,1", &p->ori);       /* seqrush.rs:49 */
    p->min_match_len = 0; p->max_divergence = -1.0; p->exclude_self = 0;
    p->memory_mode = SRO_MEM_ULTRALOW; p->threads = 4;
}

void sro_alignment_free(sro_alignment *a) { free(a->cigar_bytes); a->cigar_bytes = NULL; }

/* Orientation rule (project decision, the allwave source is absent): the
 * query is scored forward and reverse-complemented against the target with
 * the orientation penalties; reverse is chosen iff its score is strictly
 * lower (forward on ties).  The reverse score is only computed up to the
 * forward score - 1, which decides the same predicate. */
int sro_align_pair(const sro_seqrush *s, const sro_params *p, uint32_t q,
                   uint32_t t, sro_alignment *out) {
    const sro_sequence *Q = &s->seqs[q], *T = &s->seqs[t];
    memset(out, 0, sizeof(*out));
    out->query_idx = q; out->target_idx = t;
    int fwd = 0, rev = INT_MAX;
    if (sro_wfa_score(Q->data, (int)Q->len, T->data, (int)T->len, &p->ori, -1, &fwd)) return -1;
    uint8_t *rc = NULL;
    int is_rev = 0;
    if (fwd > 0) {
        rc = (uint8_t *)malloc(Q->len);
        sro_reverse_complement(Q->data, Q->len, rc);
        if (sro_wfa_score(rc, (int)Q->len, T->data, (int)T->len, &p->ori, fwd - 1, &rev)) { free(rc); return -1; }
        is_rev = rev < fwd;
    }
    const uint8_t *qq = is_rev ? rc : Q->data;
    int st = sro_wfa_align(qq, (int)Q->len, T->data, (int)T->len, &p->pen,
                           p->memory_mode, &out->cigar_bytes, &out->cigar_len, &out->score);
    free(rc);
    if (st) return st;
    out->is_reverse = is_rev;
    out->query_start = 0; out->query_end = Q->len;
    out->target_start = 0; out->target_end = T->len;
    return 0;
}

int64_t sro_align_and_unite(sro_seqrush *s, const sro_params *p,
                            uint64_t pair_begin, uint64_t pair_end,
                            uint64_t *dp_cells) {
    const uint64_t n = s->n;
    if (pair_end > n * n) pair_end = n * n;
    int64_t done = 0;
    uint64_t cells = 0;
    int failed = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(p->threads > 0 ? p->threads : 1) reduction(+:done,cells)
#endif
    for (uint64_t idx = pair_begin; idx < pair_end; idx++) {
        uint32_t q = (uint32_t)(idx / n), t = (uint32_t)(idx % n);
        if (p->exclude_self && q == t) continue;
        sro_alignment a;
        if (sro_align_pair(s, p, q, t, &a)) { failed = 1; continue; }
        cells += s->seqs[q].len * s->seqs[t].len;
        int keep = 1;
        if (p->max_divergence >= 0.0) {
            uint64_t L = s->seqs[q].len < s->seqs[t].len ? s->seqs[q].len : s->seqs[t].len;
            if (a.score > sro_max_score_for_divergence(&p->pen, L, p->max_divergence)) keep = 0;
        }
        if (keep) {
            char *cig = sro_cigar_bytes_to_string(a.cigar_bytes, a.cigar_len);  /* seqrush.rs:740 */
            int64_t r = sro_process_alignment(s, cig, q, t, p->min_match_len, a.is_reverse,
                                              0, s->seqs[q].len, 0, s->seqs[t].len); /* :744-755 */
            if (r < 0) failed = 1;
            free(cig);
        }
        sro_alignment_free(&a);
        done++;
    }
    if (dp_cells) *dp_cells = cells;
    return failed ? -1 : done;
}

/* the same over an explicit ordered pair list (sparsified lists, explicit lists) */
int64_t sro_align_and_unite_list(sro_seqrush *s, const sro_params *p, const uint32_t *pq, const uint32_t *pt,
                                 uint64_t count, uint64_t *dp_cells) {
    int64_t done = 0;
    uint64_t cells = 0;
    int failed = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(p->threads > 0 ? p->threads : 1) reduction(+:done,cells)
#endif
    for (uint64_t idx = 0; idx < count; idx++) {
        const uint32_t q = pq[idx], t = pt[idx];
        sro_alignment a;
        if (sro_align_pair(s, p, q, t, &a)) { failed = 1; continue; }
        cells += s->seqs[q].len * s->seqs[t].len;
        int keep = 1;
        if (p->max_divergence >= 0.0) {
            uint64_t L = s->seqs[q].len < s->seqs[t].len ? s->seqs[q].len : s->seqs[t].len;
            if (a.score > sro_max_score_for_divergence(&p->pen, L, p->max_divergence)) keep = 0;
        }
        if (keep) {
            char *cig = sro_cigar_bytes_to_string(a.cigar_bytes, a.cigar_len);
            int64_t r = sro_process_alignment(s, cig, q, t, p->min_match_len, a.is_reverse,
                                              0, s->seqs[q].len, 0, s->seqs[t].len);
            if (r < 0) failed = 1;
            free(cig);
        }
        sro_alignment_free(&a);
        done++;
    }
    if (dp_cells) *dp_cells = cells;
    return failed ? -1 : done;
}

/* the same, and the per-pair results kept for a full-size comparison (tests): score, strand, length and a 64-bit
 * digest of the CIGAR runs (and their count) of every pair, in list order.  Test infrastructure like everything here. */
/* digest of a raw CIGAR (bytes M X I D): its maximal runs as words (len << 4) | code, code M=0 X=1 D=2 I=3 (the
 * device's op codes), summed as mix(word ^ index * golden) mod 2^64 -- order-sensitive and computable with array
 * operations on the other side */
static uint64_t dg_mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
static uint64_t cigar_run_digest(const uint8_t *b, int64_t n, uint32_t *nruns) {
    uint64_t h = 0, idx = 0;
    int64_t i = 0;
    while (i < n) {
        int64_t j = i;
        while (j < n && b[j] == b[i]) j++;
        const uint64_t code = b[i] == 'M' ? 0 : b[i] == 'X' ? 1 : b[i] == 'D' ? 2 : 3;
        const uint64_t w = ((uint64_t)(j - i) << 4) | code;
        h += dg_mix(w ^ (idx * 0x9e3779b97f4a7c15ULL));
        idx++; i = j;
    }
    if (nruns) *nruns = (uint32_t)idx;
    return h;
}
uint64_t sro_cigar_run_digest(const uint8_t *b, uint64_t n) { return cigar_run_digest(b, (int64_t)n, 0); }
int64_t sro_align_and_unite_list_collect(sro_seqrush *s, const sro_params *p, const uint32_t *pq, const uint32_t *pt,
                                         uint64_t count, int do_unite, int32_t *score, uint8_t *is_reverse,
                                         uint32_t *cigar_len, uint64_t *cigar_digest) {
    int64_t done = 0;
    int failed = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(p->threads > 0 ? p->threads : 1) reduction(+:done)
#endif
    for (uint64_t idx = 0; idx < count; idx++) {
        const uint32_t q = pq[idx], t = pt[idx];
        sro_alignment a;
        if (sro_align_pair(s, p, q, t, &a)) { failed = 1; continue; }
        score[idx] = a.score; is_reverse[idx] = (uint8_t)(a.is_reverse != 0);
        cigar_digest[idx] = cigar_run_digest(a.cigar_bytes, a.cigar_len, &cigar_len[idx]);
        int keep = do_unite;
        if (keep && p->max_divergence >= 0.0) {
            uint64_t L = s->seqs[q].len < s->seqs[t].len ? s->seqs[q].len : s->seqs[t].len;
            if (a.score > sro_max_score_for_divergence(&p->pen, L, p->max_divergence)) keep = 0;
        }
        if (keep) {
            char *cig = sro_cigar_bytes_to_string(a.cigar_bytes, a.cigar_len);
            int64_t r = sro_process_alignment(s, cig, q, t, p->min_match_len, a.is_reverse,
                                              0, s->seqs[q].len, 0, s->seqs[t].len);
            if (r < 0) failed = 1;
            free(cig);
        }
        sro_alignment_free(&a);
        done++;
    }
    return failed ? -1 : done;
}

/* ------------------------------------------------------------------ */
/* pair sparsification (grammar seqrush.rs:356-431; the selection rules live in the absent allwave crate:    */
/* this is the project's own definition, the same one the product implements -- PARITY UNPINNED)             */
/*   random:F        ordered pair (q,t), q != t, kept iff unit(mix(seed ^ (q*n+t))) < F                         */
/*   connectivity:P  unordered {i<j} kept iff unit(mix(seed ^ (i*n+j))) < min(1, (ln n - ln(-ln P)) / n)       */
/*   auto            n < 10: none, else connectivity:0.99                                                      */
/*   tree:kn,kf,rf,k unordered pair kept iff one of the kn nearest / kf farthest neighbours of either end by   */
/*                   bottom-1000 sketch similarity of canonical k-mers, or unit(mix(seed ^ (i*n+j))) < rf      */
/* self pairs are always kept unless exclude_self; both directions of a kept unordered pair are aligned        */
/* ------------------------------------------------------------------ */
static uint64_t sp_mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
static double sp_unit(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }
static int sp_code(uint8_t b) {
    switch (b) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return -1; }
}
static int sp_cmp_u64(const void *a, const void *b) {
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
#define SP_SKETCH 1000
/* bottom-SP_SKETCH distinct canonical k-mer hashes of one sequence, ascending; returns the count */
static uint32_t sp_sketch(const sro_sequence *q, int k, uint64_t *out) {
    if (q->len < (uint64_t)k) return 0;
    const uint64_t nk = q->len - (uint64_t)k + 1;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * nk);
    uint64_t m = 0;
    const uint64_t salt = (uint64_t)k * 0x9E3779B97F4A7C15ULL;
    for (uint64_t i = 0; i < nk; i++) {
        uint64_t f = 0, r = 0;
        int ok = 1;
        for (int j = 0; j < k; j++) {
            const int c = sp_code(q->data[i + (uint64_t)j]);
            if (c < 0) { ok = 0; break; }
            f = (f << 2) | (uint64_t)c;
            r |= (uint64_t)(3 - c) << (2 * j);
        }
        if (!ok) continue;
        const uint64_t v = sp_mix((f < r ? f : r) ^ salt);
        if (v != UINT64_MAX) h[m++] = v;
    }
    qsort(h, m, sizeof(uint64_t), sp_cmp_u64);
    uint32_t cnt = 0;
    for (uint64_t i = 0; i < m && cnt < SP_SKETCH; i++)
        if (i == 0 || h[i] != h[i - 1]) out[cnt++] = h[i];
    free(h);
    return cnt;
}

int sro_sparsified_pairs(const sro_seqrush *s, const sro_sparsification *sp, uint64_t seed, int exclude_self,
                         uint32_t **pq_out, uint32_t **pt_out, uint64_t *count) {
    const uint64_t n = s->n;
    int kind = sp->kind;
    double frac = sp->factor;
    if (kind == SRO_SPARSE_AUTO) { if (n < 10) kind = SRO_SPARSE_NONE; else { kind = SRO_SPARSE_CONNECTIVITY; frac = 0.99; } }
    double conn_f = 1.0;
    if (kind == SRO_SPARSE_CONNECTIVITY && n > 2 && frac < 1.0) {
        const double c = -log(-log(frac));
        conn_f = (log((double)n) + c) / (double)n;
        if (conn_f > 1.0) conn_f = 1.0;
        if (conn_f < 0.0) conn_f = 0.0;
    }
    uint8_t *sel = NULL;
    if (kind == SRO_SPARSE_TREE && n > 1) {
        if (sp->kmer_size < 1 || sp->kmer_size > 32) return -1;
        uint64_t *sk = (uint64_t *)malloc(sizeof(uint64_t) * n * SP_SKETCH);
        uint32_t *skn = (uint32_t *)malloc(sizeof(uint32_t) * n);
        for (uint64_t i = 0; i < n; i++) skn[i] = sp_sketch(&s->seqs[i], (int)sp->kmer_size, sk + i * SP_SKETCH);
        uint32_t *sh = (uint32_t *)calloc(n * n, sizeof(uint32_t)), *dn = (uint32_t *)calloc(n * n, sizeof(uint32_t));
        for (uint64_t i = 0; i < n; i++)
            for (uint64_t j = i + 1; j < n; j++) {
                const uint64_t *A = sk + i * SP_SKETCH, *B = sk + j * SP_SKETCH;
                uint32_t x = 0, y = 0, shared = 0, denom = 0;
                while (denom < SP_SKETCH && (x < skn[i] || y < skn[j])) {
                    if (y >= skn[j] || (x < skn[i] && A[x] < B[y])) x++;
                    else if (x >= skn[i] || B[y] < A[x]) y++;
                    else { x++; y++; shared++; }
                    denom++;
                }
                if (denom == 0) denom = 1;
                sh[i * n + j] = sh[j * n + i] = shared; dn[i * n + j] = dn[j * n + i] = denom;
            }
        sel = (uint8_t *)calloc(n * n, 1);
        const uint64_t kn = sp->k_nearest < n ? sp->k_nearest : n, kf = sp->k_farthest < n ? sp->k_farthest : n;
        for (uint64_t i = 0; i < n; i++) {
            uint8_t *row = sel + i * n;
            for (uint64_t pass = 0; pass < kn + kf; pass++) {
                const int nearest = pass < kn;
                int64_t best = -1;
                for (uint64_t j = 0; j < n; j++) {
                    if (j == i || (row[j] & (nearest ? 1 : 2))) continue;
                    if (best < 0) { best = (int64_t)j; continue; }
                    const uint64_t l = (uint64_t)sh[i * n + j] * dn[i * n + (uint64_t)best];
                    const uint64_t r = (uint64_t)sh[i * n + (uint64_t)best] * dn[i * n + j];
                    if (nearest ? (l > r) : (l < r)) best = (int64_t)j;
                }
                if (best < 0) break;
                row[best] |= nearest ? 1 : 2;
            }
        }
        free(sk); free(skn); free(sh); free(dn);
    }
    const uint64_t cap = n ? n * n : 1;
    uint32_t *pq = (uint32_t *)malloc(sizeof(uint32_t) * cap), *pt = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    uint64_t m = 0;
    for (uint64_t q = 0; q < n; q++)
        for (uint64_t t = 0; t < n; t++) {
            if (q == t) { if (exclude_self) continue; pq[m] = (uint32_t)q; pt[m] = (uint32_t)t; m++; continue; }
            const uint64_t i = q < t ? q : t, j = q < t ? t : q;
            int keep = 1;
            if (kind == SRO_SPARSE_RANDOM) keep = sp_unit(sp_mix(seed ^ (q * n + t))) < frac;
            else if (kind == SRO_SPARSE_CONNECTIVITY) keep = sp_unit(sp_mix(seed ^ (i * n + j))) < conn_f;
            else if (kind == SRO_SPARSE_TREE)
                keep = (sel && (sel[i * n + j] || sel[j * n + i])) || sp_unit(sp_mix(seed ^ (i * n + j))) < sp->rand_frac;
            if (keep) { pq[m] = (uint32_t)q; pt[m] = (uint32_t)t; m++; }
        }
    free(sel);
    *pq_out = pq; *pt_out = pt; *count = m;
    return 0;
}

/* ------------------------------------------------------------------ */
/* graph induction + GFA                                                */
/* ------------------------------------------------------------------ */
void sro_canonical_labels(sro_seqrush *s, uint64_t *labels) {
    uint64_t n = sro_uf_size(s->uf);
    uint64_t *minof = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    for (uint64_t i = 0; i < n; i++) minof[i] = UINT64_MAX;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t r = sro_uf_find(s->uf, i);
        if (i < minof[r]) minof[r] = i;
    }
    for (uint64_t i = 0; i < n; i++) labels[i] = minof[sro_uf_find(s->uf, i)];
    free(minof);
}

typedef struct { char *b; size_t n, cap; } sbuf;
static void sb_reserve(sbuf *s, size_t extra) {
    if (s->n + extra + 1 > s->cap) {
        size_t nc = s->cap ? s->cap : 4096;
        while (nc < s->n + extra + 1) nc *= 2;
        s->b = (char *)realloc(s->b, nc);
        s->cap = nc;
    }
}
static void sb_puts(sbuf *s, const char *t) {
    size_t l = strlen(t);
    sb_reserve(s, l);
    memcpy(s->b + s->n, t, l); s->n += l; s->b[s->n] = 0;
}
static void sb_putu(sbuf *s, uint64_t v) {
    char t[24]; snprintf(t, sizeof(t), "%llu", (unsigned long long)v); sb_puts(s, t);
}
static void sb_putc(sbuf *s, char c) { sb_reserve(s, 1); s->b[s->n++] = c; s->b[s->n] = 0; }

/* open-addressing set of (from,to) handle pairs, keeps insertion order */
typedef struct { uint64_t *keys; uint64_t cap, n; uint64_t *order_from, *order_to; uint64_t ocap; } eset;
static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
static void eset_init(eset *e, uint64_t cap) {
    e->cap = 64; while (e->cap < cap * 2) e->cap *= 2;
    e->keys = (uint64_t *)malloc(sizeof(uint64_t) * e->cap * 2);
    for (uint64_t i = 0; i < e->cap * 2; i++) e->keys[i] = UINT64_MAX;
    e->n = 0; e->ocap = 1024;
    e->order_from = (uint64_t *)malloc(sizeof(uint64_t) * e->ocap);
    e->order_to = (uint64_t *)malloc(sizeof(uint64_t) * e->ocap);
}
static int eset_has(const eset *e, uint64_t f, uint64_t t) {
    uint64_t h = mix64(f * 0x9e3779b97f4a7c15ULL ^ t) & (e->cap - 1);
    while (e->keys[2 * h] != UINT64_MAX) {
        if (e->keys[2 * h] == f && e->keys[2 * h + 1] == t) return 1;
        h = (h + 1) & (e->cap - 1);
    }
    return 0;
}
static void eset_grow(eset *e);
static void eset_put(eset *e, uint64_t f, uint64_t t, int record) {
    if ((e->n + 1) * 2 > e->cap) eset_grow(e);
    uint64_t h = mix64(f * 0x9e3779b97f4a7c15ULL ^ t) & (e->cap - 1);
    while (e->keys[2 * h] != UINT64_MAX) h = (h + 1) & (e->cap - 1);
    e->keys[2 * h] = f; e->keys[2 * h + 1] = t;
    if (record) {
        if (e->n == e->ocap) {
            e->ocap *= 2;
            e->order_from = (uint64_t *)realloc(e->order_from, sizeof(uint64_t) * e->ocap);
            e->order_to = (uint64_t *)realloc(e->order_to, sizeof(uint64_t) * e->ocap);
        }
        e->order_from[e->n] = f; e->order_to[e->n] = t;
    }
    e->n++;
}
static void eset_grow(eset *e) {
    uint64_t *old = e->keys, oc = e->cap, n = e->n;
    e->cap *= 2;
    e->keys = (uint64_t *)malloc(sizeof(uint64_t) * e->cap * 2);
    for (uint64_t i = 0; i < e->cap * 2; i++) e->keys[i] = UINT64_MAX;
    e->n = 0;
    for (uint64_t i = 0; i < oc; i++)
        if (old[2 * i] != UINT64_MAX) eset_put(e, old[2 * i], old[2 * i + 1], 0);
    e->n = n;
    free(old);
}
static void eset_free(eset *e) { free(e->keys); free(e->order_from); free(e->order_to); }

/* find_sequence_for_position bidirected_builder.rs:328-333 */
static const sro_sequence *find_seq_for_pos(const sro_seqrush *s, sro_pos pos) {
    uint64_t off = sro_offset(pos);
    for (uint64_t i = 0; i < s->n; i++)
        if (off >= s->seqs[i].offset && off < s->seqs[i].offset + s->seqs[i].len) return &s->seqs[i];
    return NULL;
}

char *sro_build_gfa(sro_seqrush *s, int canonical, int faithful_scan,
                    uint64_t *n_nodes, uint64_t *n_edges) {
    const uint64_t ufn = sro_uf_size(s->uf);
    uint64_t *label = NULL;
    if (canonical) {
        label = (uint64_t *)malloc(sizeof(uint64_t) * ufn);
        sro_canonical_labels(s, label);
    }
#define FIND(p) (canonical ? label[(p)] : sro_buf_find(s->uf, (p)))
    /* union_to_node: HashMap<Pos,usize> (:25) -> direct table over Pos */
    uint64_t *union_to_node = (uint64_t *)calloc(ufn, sizeof(uint64_t));
    uint64_t *keys = NULL, nkeys = 0, kcap = 0;   /* insertion-ordered key list */
    uint64_t next_node_id = 1;
    uint8_t *node_base = NULL; uint64_t nbcap = 0;
    /* paths */
    uint64_t **steps = (uint64_t **)calloc(s->n ? s->n : 1, sizeof(uint64_t *));
    for (uint64_t si = 0; si < s->n; si++) {
        const sro_sequence *seq = &s->seqs[si];
        steps[si] = (uint64_t *)malloc(sizeof(uint64_t) * seq->len);
        for (uint64_t i = 0; i < seq->len; i++) {
            uint64_t global_pos = seq->offset + i;
            sro_pos pos_fwd = sro_make_pos(global_pos, 0), pos_rev = sro_make_pos(global_pos, 1);
            sro_pos union_fwd = FIND(pos_fwd), union_rev = FIND(pos_rev);   /* :48-49 */
            sro_pos union_rep;
            if (union_to_node[union_fwd]) union_rep = union_fwd;            /* :53-58 */
            else if (union_to_node[union_rev]) union_rep = union_rev;
            else {
                /* :60-127.  fwd_root/rev_root are the same finds again; then
                 * the scan over every existing key calling same(): keys are
                 * roots and union_fwd/rev are roots, so same() can only hold
                 * for equal keys, which the two lookups above excluded.  The
                 * literal scan is kept behind faithful_scan for small cases. */
                sro_pos found = UINT64_MAX;
                if (faithful_scan) {
                    for (uint64_t q = 0; q < nkeys; q++) {
                        uint64_t ex = keys[q];
                        if (FIND(union_fwd) == FIND(ex) || FIND(union_rev) == FIND(ex)) { found = ex; break; }
                    }
                }
                union_rep = found != UINT64_MAX ? found : union_fwd;        /* :121-126 */
            }
            uint64_t node_id; uint8_t nbase;
            if (union_to_node[union_rep]) {                                 /* :145-153 */
                node_id = union_to_node[union_rep];
                nbase = node_base[node_id];
            } else {                                                        /* :154-186 */
                node_id = next_node_id++;
                union_to_node[union_rep] = node_id;
                if (nkeys == kcap) { kcap = kcap ? kcap * 2 : 1024; keys = (uint64_t *)realloc(keys, sizeof(uint64_t) * kcap); }
                keys[nkeys++] = union_rep;
                if (FIND(pos_fwd) == FIND(union_rep) && !union_to_node[union_fwd]) {
                    union_to_node[union_fwd] = node_id;
                    if (nkeys == kcap) { kcap *= 2; keys = (uint64_t *)realloc(keys, sizeof(uint64_t) * kcap); }
                    keys[nkeys++] = union_fwd;
                }
                if (FIND(pos_rev) == FIND(union_rep) && !union_to_node[union_rev]) {
                    union_to_node[union_rev] = node_id;
                    if (nkeys == kcap) { kcap *= 2; keys = (uint64_t *)realloc(keys, sizeof(uint64_t) * kcap); }
                    keys[nkeys++] = union_rev;
                }
                const sro_sequence *src = find_seq_for_pos(s, union_rep);   /* :176-182 */
                nbase = src ? src->data[sro_offset(union_rep) - src->offset] : seq->data[i];
                if (node_id >= nbcap) { nbcap = nbcap ? nbcap * 2 : 1024; while (node_id >= nbcap) nbcap *= 2; node_base = (uint8_t *)realloc(node_base, nbcap); }
                node_base[node_id] = nbase;
            }
            uint8_t expected = (uint8_t)toupper(seq->data[i]);              /* :190-199 */
            uint8_t nb = (uint8_t)toupper(nbase);
            int need_reverse = (nb == 'A' && expected == 'T') || (nb == 'T' && expected == 'A') ||
                               (nb == 'C' && expected == 'G') || (nb == 'G' && expected == 'C');
            steps[si][i] = (node_id << 1) | (uint64_t)need_reverse;         /* Handle::new */
        }
    }
    /* edges from consecutive steps :217-228, add_edge bidirected_ops.rs:813-825 */
    eset es; eset_init(&es, s->total_length);
    for (uint64_t si = 0; si < s->n; si++) {
        const uint64_t L = s->seqs[si].len;
        for (uint64_t i = 0; i + 1 < L; i++) {
            uint64_t from = steps[si][i], to = steps[si][i + 1];
            uint64_t cf = to ^ 1, ct = from ^ 1;                            /* complement */
            if (!eset_has(&es, from, to) && !eset_has(&es, cf, ct)) eset_put(&es, from, to, 1);
        }
    }
    /* write_gfa bidirected_ops.rs:880-925 */
    sbuf out = {0, 0, 0};
    sb_puts(&out, "H\tVN:Z:1.0\n");
    for (uint64_t id = 1; id < next_node_id; id++) {
        sb_puts(&out, "S\t"); sb_putu(&out, id); sb_putc(&out, '\t');
        sb_putc(&out, (char)node_base[id]); sb_putc(&out, '\n');
    }
    for (uint64_t i = 0; i < es.n; i++) {
        uint64_t f = es.order_from[i], t = es.order_to[i];
        sb_puts(&out, "L\t"); sb_putu(&out, f >> 1); sb_putc(&out, '\t');
        sb_putc(&out, (f & 1) ? '-' : '+'); sb_putc(&out, '\t');
        sb_putu(&out, t >> 1); sb_putc(&out, '\t');
        sb_putc(&out, (t & 1) ? '-' : '+'); sb_puts(&out, "\t0M\n");
    }
    for (uint64_t si = 0; si < s->n; si++) {
        sb_puts(&out, "P\t"); sb_puts(&out, s->seqs[si].id); sb_putc(&out, '\t');
        for (uint64_t i = 0; i < s->seqs[si].len; i++) {
            if (i) sb_putc(&out, ',');
            sb_putu(&out, steps[si][i] >> 1);
            sb_putc(&out, (steps[si][i] & 1) ? '-' : '+');
        }
        sb_puts(&out, "\t*\n");
    }
    if (n_nodes) *n_nodes = next_node_id - 1;
    if (n_edges) *n_edges = es.n;
    eset_free(&es);
    for (uint64_t si = 0; si < s->n; si++) free(steps[si]);
    free(steps); free(union_to_node); free(keys); free(node_base); free(label);
#undef FIND
    return out.b;
}
