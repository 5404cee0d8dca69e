// seqrush_cli.cpp -- C++ host side above the C ABI: the reference's CLI surface for the hot path
// (src/main.rs:4-7, Args src/seqrush.rs:17-152, run_seqrush :1839-1853, load_sequences :1801-1837).
// Everything that computes goes through include/seqrush_amd.h; output is the --no-sort graph, compacted unless
// --no-compact (the Ygs layout is outside the hot path).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "../../include/seqrush_amd.h"

struct Seq { std::string id; std::string data; };

static bool is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c; }

// load_sequences, src/seqrush.rs:1801-1837
static bool load_sequences(const std::string &path, std::vector<Seq> &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string line, cur_id, cur;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            if (!cur_id.empty()) { out.push_back({cur_id, cur}); cur.clear(); }   // data kept when the id was empty (:1812-1820)
            size_t a = 1;
            while (a < line.size() && is_ws((unsigned char)line[a])) a++;
            size_t b = a;
            while (b < line.size() && !is_ws((unsigned char)line[b])) b++;
            cur_id = line.substr(a, b - a);
        } else {
            size_t a = 0, b = line.size();
            while (a < b && is_ws((unsigned char)line[a])) a++;
            while (b > a && is_ws((unsigned char)line[b - 1])) b--;
            cur.append(line, a, b - a);
        }
    }
    if (!cur_id.empty()) out.push_back({cur_id, cur});
    return true;
}

// label part file of a shard run: header + uf_size canonical labels (u64).  The header lets the merge run refuse parts of
// another input, another shard count, a duplicated part or an incomplete set.
struct PartHeader { char magic[8]; uint64_t uf_size, shard_rank, shard_count, input_hash, flags; };
static const char PART_MAGIC[8] = {'S', 'R', 'L', 'A', 'B', 'E', 'L', '1'};
static uint64_t fnv1a(const void *data, size_t n, uint64_t h = 0xcbf29ce484222325ULL) {
    const unsigned char *b = (const unsigned char *)data;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
    return h;
}

static void usage() {
    fprintf(stderr, "usage: seqrush_mi355x -s in.fa [-o output.gfa] [-k 0] [-S 0,5,8,2,24,1] [--orientation-scores 0,1,1,1]\n"
                    "       [-d max_divergence] [-x none|auto|random:F|connectivity:P|tree:kn[,kf[,rf[,k]]]] [-p in.paf] [--output-alignments out.paf] --no-sort [--no-compact] [--device N]\n"
                    "       [--shard R/N --labels-out part.bin]  |  [--labels-in part0.bin --labels-in part1.bin ...]\n");
}

int main(int argc, char **argv) {
    std::string sequences, output = "output.gfa", scores = "0,5,8,2,24,1", ori = "0,1,1,1", sparsify = "none", paf_out, paf_in,
                aligner = "allwave";
    long long k = 0;
    double max_div = -1.0;
    int device = 0;
    bool no_sort = false, no_compact = false;
    // multi-GPU without a collective library in this host: every process aligns one shard (--shard R/N) and writes its
    // canonical labels (--labels-out); a last run merges the files (--labels-in, repeatable) and writes the graph.
    // (With RCCL at hand the exchange is one all-gather: python -m seqrush_amd --gpus N, bench.py.)
    unsigned shard_rank = 0, shard_count = 1;
    std::string labels_out;
    std::vector<std::string> labels_in;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "error: %s needs a value\n", name); exit(2); }
            return argv[++i];
        };
        if (a == "-s" || a == "--sequences") sequences = val("-s");
        else if (a == "-o" || a == "--output") output = val("-o");
        else if (a == "-k" || a == "--min-match-length") k = atoll(val("-k"));
        else if (a == "-t" || a == "--threads") (void)val("-t");               // host threads: unused by the device path
        else if (a == "-S" || a == "--scores") scores = val("-S");
        else if (a == "--orientation-scores") ori = val("--orientation-scores");
        else if (a == "-d" || a == "--max-divergence") max_div = atof(val("-d"));
        else if (a == "-x" || a == "--sparsify") sparsify = val("-x");
        else if (a == "-p" || a == "--paf") paf_in = val("-p");
        else if (a == "--output-alignments") paf_out = val("--output-alignments");
        else if (a == "--aligner") aligner = val("--aligner");
        else if (a == "--no-sort") no_sort = true;
        else if (a == "--no-compact") no_compact = true;
        else if (a == "--device") device = atoi(val("--device"));
        else if (a == "--shard") { if (sscanf(val("--shard"), "%u/%u", &shard_rank, &shard_count) != 2 || shard_count == 0 || shard_rank >= shard_count) { fprintf(stderr, "error: --shard R/N\n"); return 2; } }
        else if (a == "--labels-out") labels_out = val("--labels-out");
        else if (a == "--labels-in") labels_in.push_back(val("--labels-in"));
        else if (a == "-v" || a == "--verbose") {}
        else { usage(); return 2; }
    }
    if (sequences.empty()) { usage(); return 2; }
    if (shard_count > 1 && labels_out.empty()) {
        fprintf(stderr, "Error: --shard %u/%u aligns a part of the pair list only: give --labels-out and merge the parts with --labels-in "
                        "(a graph of one shard would be silently incomplete)\n", shard_rank, shard_count);
        return 1;
    }
    if (!labels_in.empty() && (shard_count > 1 || !labels_out.empty())) { fprintf(stderr, "Error: --labels-in is the merge run: no --shard / --labels-out\n"); return 1; }
    if (aligner != "allwave" && aligner != "AllWave") { fprintf(stderr, "Error: aligner '%s' is out of scope; only 'allwave'\n", aligner.c_str()); return 1; }
    if (!no_sort) { fprintf(stderr, "Error: only --no-sort output is implemented (the Ygs layout is outside the hot path); compaction runs unless --no-compact\n"); return 1; }
    std::vector<Seq> seqs;
    if (!load_sequences(sequences, seqs)) { fprintf(stderr, "Error: cannot read %s\n", sequences.c_str()); return 1; }
    printf("Loaded %zu sequences\n", seqs.size());
    std::string bases;
    std::vector<uint64_t> offsets(1, 0);
    std::vector<const char *> names;
    for (auto &s : seqs) { bases += s.data; offsets.push_back(bases.size()); names.push_back(s.id.c_str()); }
    sr_seqset set{(uint32_t)seqs.size(), (const uint8_t *)bases.data(), offsets.data(), names.data()};
    sr_params p;
    sr_default_params(&p);
    if (sr_parse_scores(scores.c_str(), &p) || sr_parse_orientation_scores(ori.c_str(), &p) ||
        sr_parse_sparsification(sparsify.c_str(), &p)) {
        fprintf(stderr, "Error: %s\n", sr_last_error());
        return 1;
    }
    p.min_match_len = (uint64_t)k; p.max_divergence = max_div; p.device = device; p.canonical_labels = 1;
    p.shard_rank = shard_rank; p.shard_count = shard_count;
    // what a part file must agree on: the sequences (bytes and boundaries) and everything that decides pairs and unions
    uint64_t input_hash = fnv1a(bases.data(), bases.size());
    input_hash = fnv1a(offsets.data(), offsets.size() * 8, input_hash);
    {
        const std::string cfg = scores + "|" + ori + "|" + sparsify + "|" + std::to_string(k) + "|" + std::to_string(max_div);
        input_hash = fnv1a(cfg.data(), cfg.size(), input_hash);
    }
    printf("Building graph with %zu sequences (total length: %zu)\n", seqs.size(), bases.size());
    printf("Total sequence pairs: %zu (sparsification: %s)\n", seqs.size() * seqs.size(), sparsify.c_str());
    // one resident context: load (or PAF replay) -> align -> unite -> graph induction, all on the device
    sr_ctx *ctx = nullptr;
    auto die = [&]() { fprintf(stderr, "Error: %s\n", sr_last_error()); if (ctx) sr_ctx_destroy(ctx); return 1; };
    if (sr_ctx_create(device, &ctx)) return die();
    if (!labels_in.empty()) {                                // merge run: no pairs of its own, the forests come from files
        if (sr_ctx_load_pairs(ctx, &set, &p, nullptr, nullptr, 0)) return die();
        const uint64_t ufn = sr_ctx_uf_size(ctx);
        std::vector<uint64_t> lab(ufn);
        std::vector<char> seen;
        uint64_t parts_of = 0;
        for (const std::string &path : labels_in) {
            FILE *f = fopen(path.c_str(), "rb");
            PartHeader h;
            auto bad = [&](const char *why) { fprintf(stderr, "Error: label part %s: %s\n", path.c_str(), why); if (f) fclose(f); sr_ctx_destroy(ctx); return 1; };
            if (!f || fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, PART_MAGIC, 8) != 0) return bad("not a label part file (no header)");
            if (h.uf_size != ufn || h.input_hash != input_hash) return bad("written for other sequences or other options (-S / --orientation-scores / -x / -k / -d)");
            if (h.shard_count == 0 || h.shard_rank >= h.shard_count) return bad("bad shard in header");
            if (parts_of == 0) { parts_of = h.shard_count; seen.assign(parts_of, 0); }
            if (h.shard_count != parts_of) return bad("belongs to a run with another shard count");
            if (seen[h.shard_rank]) return bad("shard given twice");
            seen[h.shard_rank] = 1;
            if (fread(lab.data(), 8, ufn, f) != ufn || fgetc(f) != EOF) return bad("truncated or oversized");
            fclose(f); f = nullptr;
            if (sr_ctx_merge_labels_host(ctx, lab.data(), 1)) return die();
        }
        for (uint64_t r = 0; r < parts_of; r++)
            if (!seen[r]) { fprintf(stderr, "Error: shard %llu/%llu is missing from the --labels-in set\n", (unsigned long long)r, (unsigned long long)parts_of); sr_ctx_destroy(ctx); return 1; }
    } else if (!paf_in.empty()) {                            // align_and_unite_from_paf (src/seqrush.rs:510-609)
        printf("Reading alignments from PAF file: %s\n", paf_in.c_str());
        if (sr_ctx_load_paf(ctx, &set, &p, paf_in.c_str())) return die();
    } else {
        if (sr_ctx_load(ctx, &set, &p)) return die();
    }
    if (!labels_in.empty()) {
    } else if (!paf_in.empty() || paf_out.empty()) {
        if (sr_ctx_run(ctx)) return die();                   // align + unite, batch after batch (PAF input: unite only)
    } else {                                                 // --output-alignments (src/seqrush.rs:678-716)
        sr_alignments *al = nullptr;
        if (sr_ctx_align_all(ctx, 1, &al)) return die();
        printf("Writing alignments to %s\n", paf_out.c_str());
        if (sr_write_paf(al, &set, paf_out.c_str())) { sr_alignments_free(al); return die(); }
        sr_alignments_free(al);
    }
    if (sr_ctx_sync(ctx)) return die();
    if (!labels_out.empty()) {                               // shard run: the forest's canonical labels, no graph
        const uint64_t ufn = sr_ctx_uf_size(ctx);
        std::vector<uint64_t> lab(ufn);
        if (sr_ctx_download_labels(ctx, lab.data())) return die();
        FILE *f = fopen(labels_out.c_str(), "wb");
        PartHeader h;
        memcpy(h.magic, PART_MAGIC, 8); h.uf_size = ufn; h.shard_rank = shard_rank; h.shard_count = shard_count; h.input_hash = input_hash; h.flags = 0;
        if (!f || fwrite(&h, sizeof(h), 1, f) != 1 || fwrite(lab.data(), 8, ufn, f) != ufn) { fprintf(stderr, "Error: cannot write %s\n", labels_out.c_str()); if (f) fclose(f); sr_ctx_destroy(ctx); return 1; }
        fclose(f);
        printf("Labels of shard %u/%u written to %s\n", shard_rank, shard_count, labels_out.c_str());
        sr_ctx_destroy(ctx);
        return 0;
    }
    char *gfa = nullptr;
    uint64_t nn = 0, ne = 0;
    if (sr_ctx_build_gfa_opts(ctx, &set, no_compact ? 0 : 1, &gfa, &nn, &ne)) return die();   // compact + renumber unless --no-compact
    sr_ctx_destroy(ctx);
    std::ofstream o(output, std::ios::binary);
    o << gfa;
    sr_free(gfa);
    printf("Graph written to %s\n", output.c_str());
    return 0;
}
