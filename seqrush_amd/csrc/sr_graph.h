// sr_graph.h -- host-side container of the induced graph (BidirectedGraph, src/bidirected_ops.rs:8-13) between
// graph induction (device: sr_graph.hip, host: sr_build_gfa), compaction (sr_compact.cpp) and the GFA writer.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

struct SrGraph {
    std::vector<std::string> node_seq;               // [node id]; slot 0 is never used (ids start at 1, ops:804-810)
    std::vector<uint8_t> node_alive;
    std::vector<uint32_t> steps;                     // all paths back to back: handle = node_id << 1 | is_reverse
    std::vector<uint64_t> path_off;                  // [npaths + 1]
    std::vector<std::pair<uint32_t, uint32_t>> edges;   // (from, to) handles, first-seen order and orientation
};

void sr_graph_compact(SrGraph &g);                   // BidirectedGraph::compact, ops:91-112
void sr_graph_renumber(SrGraph &g);                  // renumber_nodes_sequentially, ops:75-89
char *sr_graph_format_gfa(const SrGraph &g, const char *const *names, uint64_t *n_nodes, uint64_t *n_edges);   // write_gfa, ops:880-925
