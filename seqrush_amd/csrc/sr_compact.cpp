// sr_compact.cpp -- linear-chain compaction + sequential renumbering of the induced graph:
// BidirectedGraph::compact (src/bidirected_ops.rs:91-112), find_simple_components (:115-275),
// merge_component_v2 (:279-490), renumber_nodes_sequentially (:75-89), as write_bidirected_gfa runs them for
// --no-sort without --no-compact (src/bidirected_gfa_writer.rs:39-51).
//
// Same results as the reference's procedure -- including the order in which its greedy, forward-only chain search
// finds chains (that order fixes the ids of the merged nodes), its multi-round fixpoint, and its refusal to merge a
// chain that some path enters in the middle -- but every predicate is answered from per-round tables instead of a
// rescan of all paths:
//   * out / in degree over the bidirected edge set (each stored edge also counts as its implied reverse, :126-143);
//   * are_perfect_neighbors(from, to) (:146-207): every occurrence of `from` is followed by `to` and every occurrence
//     of to.flip() by from.flip()  <=>  one pass over the steps that records, per handle, its visits, its first
//     successor and whether any occurrence disagrees or ends a path;
//   * merge validation (:318-363): the reference walks every path and demands that any step belonging to the chain
//     starts a complete forward or reverse traversal; here only the occurrences of the chain's handles are visited,
//     in path order, with the same skip-ahead;
//   * all chains of a round are node-disjoint, so their path and edge rewrites (:369-477) are applied in one pass.
// One round is O(steps + edges + nodes); the reference's is O(candidates x steps).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>
#include "sr_graph.h"

typedef uint32_t hnd;                                // Handle: node_id << 1 | is_reverse
static inline uint32_t hid(hnd h) { return h >> 1; }

static inline uint8_t rc_node_base(uint8_t b) {      // src/bidirected_graph.rs:73-85
    switch (b) {
    case 'A': case 'a': return 'T'; case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C';
    case 'N': case 'n': return 'N';
    default: return b;
    }
}

struct EdgeHash {
    size_t operator()(uint64_t k) const { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; return (size_t)k; }
};

// one round of compact(): returns true when at least one chain was merged
static bool compact_round(SrGraph &g) {
    const size_t NN = g.node_seq.size();             // nodes.len()
    const size_t NH = NN * 2;
    const uint32_t NONE = 0xffffffffu;
    // ---- degrees (ops:126-143)
    std::vector<uint32_t> fcnt(NH, 0), bcnt(NH, 0), fonly(NH, NONE);
    for (const auto &e : g.edges) {
        fcnt[e.first]++; fonly[e.first] = e.second; bcnt[e.second]++;
        fcnt[e.second ^ 1]++; fonly[e.second ^ 1] = e.first ^ 1; bcnt[e.first ^ 1]++;
    }
    // ---- per-handle successor table (are_perfect_neighbors, ops:146-207) + occurrence index
    const size_t NS = g.steps.size();
    std::vector<uint32_t> visits(NH, 0), succ(NH, NONE);
    std::vector<uint8_t> bad(NH, 0);
    const size_t npaths = g.path_off.size() - 1;
    for (size_t p = 0; p < npaths; p++)
        for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; i++) {
            const hnd h = g.steps[i];
            if (i + 1 < g.path_off[p + 1]) {
                const hnd s = g.steps[i + 1];
                if (visits[h] == 0) succ[h] = s; else if (succ[h] != s) bad[h] = 1;
            } else bad[h] = 1;                       // the path ends at h
            visits[h]++;
        }
    auto perfect = [&](hnd from, hnd to) {
        if (visits[from] && (bad[from] || succ[from] != to)) return false;
        const hnd tr = to ^ 1, fr = from ^ 1;
        if (visits[tr] && (bad[tr] || succ[tr] != fr)) return false;
        return true;
    };
    // ---- find_simple_components (ops:210-272), literally
    std::vector<uint8_t> visited(NH, 0), merged(NN, 0);
    std::vector<std::vector<hnd>> comps;
    for (size_t id = 0; id < NN; id++) {
        if (!g.node_alive[id]) continue;
        for (int rev = 0; rev < 2; rev++) {
            const hnd handle = (hnd)((id << 1) | (size_t)rev);
            if (visited[handle]) continue;
            if (fcnt[handle] != 1) continue;
            std::vector<hnd> chain(1, handle);
            visited[handle] = 1;
            hnd current = handle;
            for (;;) {
                if (fcnt[current] != 1) break;
                const hnd next = fonly[current];
                if (bcnt[next] != 1 || visited[next]) break;
                if (!perfect(current, next)) break;
                chain.push_back(next); visited[next] = 1; current = next;
                if (fcnt[next] != 1) break;
            }
            if (chain.size() >= 2) {
                bool already = false;
                for (hnd h : chain) if (merged[hid(h)]) { already = true; break; }
                if (!already) {
                    for (hnd h : chain) merged[hid(h)] = 1;
                    comps.push_back(std::move(chain));
                }
            }
        }
    }
    if (comps.empty()) return false;
    // ---- occurrence index of the handles that belong to some chain
    std::vector<uint64_t> occ_off(NH + 1, 0);
    {
        std::vector<uint8_t> wanted(NH, 0);
        for (const auto &c : comps) for (hnd h : c) { wanted[h] = 1; wanted[h ^ 1] = 1; }
        for (size_t i = 0; i < NS; i++) if (wanted[g.steps[i]]) occ_off[g.steps[i] + 1]++;
        for (size_t h = 0; h < NH; h++) occ_off[h + 1] += occ_off[h];
    }
    std::vector<uint64_t> occ(occ_off[NH]);
    {
        std::vector<uint64_t> fill(occ_off.begin(), occ_off.end() - 1);
        for (size_t i = 0; i < NS; i++) {
            const hnd h = g.steps[i];
            if (occ_off[h + 1] > occ_off[h]) occ[fill[h]++] = i;
        }
    }
    // end of the path a step lies in: binary search over path_off
    auto path_end = [&](uint64_t i) { return *std::upper_bound(g.path_off.begin(), g.path_off.end(), i); };
    // ---- merge_component_v2 validation (ops:318-363) per chain, then batch rewrite
    std::vector<int32_t> chain_of(NH, -1);           // validated chains: handle -> chain index (both orientations)
    std::vector<uint32_t> new_id_of;                 // per component: new node id or NONE
    uint32_t next_id = (uint32_t)NN;
    bool any = false;
    new_id_of.assign(comps.size(), NONE);
    std::vector<uint64_t> positions;
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const std::vector<hnd> &H = comps[ci];
        const size_t n = H.size();
        // handle_mapping with HashMap insert order (ops:307-315): a later insert of the same key wins
        std::vector<std::pair<hnd, uint32_t>> mp;
        mp.reserve(2 * n);
        for (size_t i = 0; i < n; i++) { mp.push_back({H[i], (uint32_t)i}); mp.push_back({(hnd)(H[i] ^ 1), (uint32_t)(n - 1 - i)}); }
        auto lookup = [&](hnd key, uint32_t *pos) {
            bool f = false;
            // chains are short on average; a repeated key can only come from a chain that revisits a node
            for (const auto &kv : mp) if (kv.first == key) { *pos = kv.second; f = true; }
            return f;
        };
        // positions of all mapped keys, in path order (distinct keys only)
        positions.clear();
        {
            std::vector<hnd> keys;
            keys.reserve(2 * n);
            for (const auto &kv : mp) keys.push_back(kv.first);
            std::sort(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
            for (hnd k : keys) for (uint64_t q = occ_off[k]; q < occ_off[k + 1]; q++) positions.push_back(occ[q]);
            std::sort(positions.begin(), positions.end());
        }
        bool ok = true;
        uint64_t covered_end = 0;
        // for long chains the per-key linear lookup would be quadratic: the chain position of a key is its index
        // (forward) or n-1-index (reverse); precompute through a small open table when n is large
        std::vector<std::pair<hnd, uint32_t>> sorted_mp;
        if (n > 16) {
            sorted_mp = mp;                           // stable: the LAST entry of equal keys must win
            std::stable_sort(sorted_mp.begin(), sorted_mp.end(), [](const std::pair<hnd, uint32_t> &a, const std::pair<hnd, uint32_t> &b) { return a.first < b.first; });
        }
        auto lookup_fast = [&](hnd key, uint32_t *pos) {
            if (n <= 16) return lookup(key, pos);
            auto it = std::upper_bound(sorted_mp.begin(), sorted_mp.end(), key, [](hnd k, const std::pair<hnd, uint32_t> &e) { return k < e.first; });
            if (it == sorted_mp.begin()) return false;
            --it;
            if (it->first != key) return false;
            *pos = it->second;
            return true;
        };
        for (uint64_t gi : positions) {
            if (gi < covered_end) continue;
            const uint64_t pe = path_end(gi);
            uint32_t cpos = 0;
            if (!lookup_fast(g.steps[gi], &cpos)) continue;            // (cannot happen: gi is an occurrence of a key)
            bool done = false;
            if (cpos == 0 && gi + n <= pe) {
                bool complete = true;
                for (size_t j = 0; j < n; j++) if (g.steps[gi + j] != H[j]) { complete = false; break; }
                if (complete) { covered_end = gi + n; done = true; }
            }
            if (!done && g.steps[gi] == (hnd)(H[n - 1] ^ 1) && gi + n <= pe) {
                bool complete = true;
                for (size_t j = 0; j < n; j++) if (g.steps[gi + j] != (hnd)(H[n - 1 - j] ^ 1)) { complete = false; break; }
                if (complete) { covered_end = gi + n; done = true; }
            }
            if (!done) { ok = false; break; }
        }
        if (!ok) continue;
        new_id_of[ci] = next_id++;
        any = true;
        for (hnd h : H) { chain_of[h] = (int32_t)ci; chain_of[h ^ 1] = (int32_t)ci; }
    }
    if (!any) return false;
    // ---- new nodes (ops:286-298, 365) in the order the merges succeeded
    g.node_seq.resize(next_id); g.node_alive.resize(next_id, 0);
    for (size_t ci = 0; ci < comps.size(); ci++) {
        if (new_id_of[ci] == NONE) continue;
        std::string s;
        for (hnd h : comps[ci]) {
            const std::string &t = g.node_seq[hid(h)];
            if (h & 1) for (size_t k = t.size(); k-- > 0;) s.push_back((char)rc_node_base((uint8_t)t[k]));
            else s += t;
        }
        g.node_seq[new_id_of[ci]] = std::move(s);
        g.node_alive[new_id_of[ci]] = 1;
    }
    // ---- path rewrite (ops:369-413), one pass for all merged chains
    {
        std::vector<hnd> ns;
        ns.reserve(NS);
        std::vector<uint64_t> noff(npaths + 1, 0);
        for (size_t p = 0; p < npaths; p++) {
            uint64_t i = g.path_off[p];
            const uint64_t pe = g.path_off[p + 1];
            while (i < pe) {
                const hnd h = g.steps[i];
                const int32_t ci = chain_of[h];
                if (ci >= 0) {
                    const std::vector<hnd> &H = comps[(size_t)ci];
                    const size_t n = H.size();
                    if (i + n <= pe) {
                        bool fwd = true;
                        for (size_t j = 0; j < n; j++) if (g.steps[i + j] != H[j]) { fwd = false; break; }
                        if (fwd) { ns.push_back((hnd)(new_id_of[(size_t)ci] << 1)); i += n; continue; }
                        bool rv = true;
                        for (size_t j = 0; j < n; j++) if (g.steps[i + j] != (hnd)(H[n - 1 - j] ^ 1)) { rv = false; break; }
                        if (rv) { ns.push_back((hnd)((new_id_of[(size_t)ci] << 1) | 1u)); i += n; continue; }
                    }
                }
                ns.push_back(h); i++;
            }
            noff[p + 1] = ns.size();
        }
        g.steps.swap(ns); g.path_off.swap(noff);
    }
    // ---- edge rewrite (ops:416-477)
    {
        std::vector<std::pair<hnd, hnd>> ne;
        ne.reserve(g.edges.size());
        std::unordered_set<uint64_t, EdgeHash> seen;
        seen.reserve(g.edges.size() * 2);
        for (const auto &e : g.edges) {
            const int32_t cf = chain_of[e.first], ct = chain_of[e.second];
            if (cf >= 0 && cf == ct) continue;                                   // internal edge of one chain
            hnd f[2], t[2];
            int nf = 0, nt = 0;
            if (cf < 0) f[nf++] = e.first;
            else {
                const std::vector<hnd> &H = comps[(size_t)cf];
                if (e.first == H.back()) f[nf++] = (hnd)(new_id_of[(size_t)cf] << 1);
                if (e.first == (hnd)(H.front() ^ 1)) f[nf++] = (hnd)((new_id_of[(size_t)cf] << 1) | 1u);
            }
            if (ct < 0) t[nt++] = e.second;
            else {
                const std::vector<hnd> &H = comps[(size_t)ct];
                if (e.second == H.front()) t[nt++] = (hnd)(new_id_of[(size_t)ct] << 1);
                if (e.second == (hnd)(H.back() ^ 1)) t[nt++] = (hnd)((new_id_of[(size_t)ct] << 1) | 1u);
            }
            for (int a = 0; a < nf; a++)
                for (int b = 0; b < nt; b++)
                    if (seen.insert(((uint64_t)f[a] << 32) | t[b]).second) ne.push_back({f[a], t[b]});
        }
        g.edges.swap(ne);
    }
    // ---- old nodes out (ops:480-487)
    for (size_t ci = 0; ci < comps.size(); ci++)
        if (new_id_of[ci] != NONE)
            for (hnd h : comps[ci]) { g.node_alive[hid(h)] = 0; std::string().swap(g.node_seq[hid(h)]); }
    return true;
}

void sr_graph_compact(SrGraph &g) {                  // compact() ops:91-112
    while (compact_round(g)) {}
}

void sr_graph_renumber(SrGraph &g) {                 // renumber_nodes_sequentially ops:75-89 + apply_node_id_mapping :21-70
    const size_t NN = g.node_seq.size();
    std::vector<uint32_t> map(NN, 0);
    uint32_t next = 1;
    for (size_t id = 0; id < NN; id++) if (g.node_alive[id]) map[id] = next++;
    std::vector<std::string> ns(next);
    std::vector<uint8_t> na(next, 0);
    for (size_t id = 0; id < NN; id++) if (g.node_alive[id]) { ns[map[id]] = std::move(g.node_seq[id]); na[map[id]] = 1; }
    g.node_seq.swap(ns); g.node_alive.swap(na);
    for (auto &e : g.edges) { e.first = (map[hid(e.first)] << 1) | (e.first & 1u); e.second = (map[hid(e.second)] << 1) | (e.second & 1u); }
    for (auto &h : g.steps) h = (map[hid(h)] << 1) | (h & 1u);
}

// write_gfa, src/bidirected_ops.rs:880-925
// decimal digits of v appended to out (the writer emits one number per node, two per edge and one per path step:
// hundreds of thousands for C2, snprintf was most of the formatting time)
static inline void put_u64(std::string &out, uint64_t v) {
    char b[20];
    int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) out += b[--n];
}
char *sr_graph_format_gfa(const SrGraph &g, const char *const *names, uint64_t *n_nodes, uint64_t *n_edges) {
    std::string out;
    size_t est = 64;
    for (const auto &s : g.node_seq) est += s.size() + 16;
    est += g.edges.size() * 32 + g.steps.size() * 9;
    out.reserve(est);
    out += "H\tVN:Z:1.0\n";
    uint64_t live = 0;
    for (size_t id = 0; id < g.node_seq.size(); id++) {
        if (!g.node_alive[id]) continue;
        live++;
        out += "S\t"; put_u64(out, id); out += '\t'; out += g.node_seq[id]; out += '\n';
    }
    for (const auto &e : g.edges) {
        out += "L\t"; put_u64(out, e.first >> 1); out += (e.first & 1) ? "\t-\t" : "\t+\t";
        put_u64(out, e.second >> 1); out += (e.second & 1) ? "\t-\t0M\n" : "\t+\t0M\n";
    }
    for (size_t p = 0; p + 1 < g.path_off.size(); p++) {
        out += "P\t"; out += names[p]; out += "\t";
        for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; i++) {
            if (i != g.path_off[p]) out += ',';
            put_u64(out, g.steps[i] >> 1); out += (g.steps[i] & 1) ? '-' : '+';
        }
        out += "\t*\n";
    }
    char *res = (char *)malloc(out.size() + 1);
    memcpy(res, out.c_str(), out.size() + 1);
    if (n_nodes) *n_nodes = live;
    if (n_edges) *n_edges = g.edges.size();
    return res;
}
