// TEST HARNESS ONLY (never shipped, never loaded by the product): compiles the exact per-line device routine
// (svjg::slow_line) and the graph-table lookups of svjedi-graph_amd/csrc/svjg_line.h with g++ and drives them
// sequentially, so their logic can be checked against the oracle on a machine without a GPU (and under
// -fsanitize=address,undefined).  The main kernel (k_classify_main) is wave/block-level code and is covered
// by the -m gpu tests through the C ABI.
#define SVJG_HD inline
#include "../../svjedi-graph_amd/csrc/svjg_line.h"
#include "../../svjedi-graph_amd/csrc/svjg_host_tables.h"
#include <vector>

using namespace svjg;

struct CountEmit {
    uint32_t *counts;
    void operator()(uint32_t slot, uint32_t allele) { counts[slot * 2 + allele]++; }
};

static GraphView make_view(const svjg_graph *g, const std::vector<uint32_t> &hash, const BucketTable &bt) {
    GraphView v;
    v.nodes = g->nodes; v.n_nodes = (uint32_t)g->n_nodes; v.edges = g->edges; v.hits = g->hits;
    v.chrom_names = (const uint8_t *)g->chrom_names; v.chrom_off = g->chrom_off; v.chrom_lo = g->chrom_node_lo;
    v.chrom_hash = hash.data(); v.n_chrom = g->n_chrom; v.hash_mask = (uint32_t)hash.size() - 1; v.d_over = g->d_over;
    v.bkt_base = bt.base.data(); v.bkt = bt.table.data(); v.bkt_shift = bt.shift;
    v.chrom_w4 = nullptr; v.chrom_wtab = nullptr; v.wtab_mask = 0;       // main-kernel dictionary: not used by the exact path
    return v;
}

extern "C" int hostsim_classify(const svjg_graph *g, const char *gaf, uint64_t n,
                                uint32_t *counts, uint64_t *n_lines, int *exc, uint64_t *err_off)
{
    std::vector<uint32_t> hash = build_chrom_hash(*g);
    BucketTable bt = build_buckets(*g);
    GraphView v = make_view(g, hash, bt);
    const uint8_t *t = (const uint8_t *)gaf;
    *n_lines = 0; *exc = 0; *err_off = 0;
    uint64_t pos = 0;
    while (pos < n) {
        uint64_t e = pos;
        while (e < n && t[e] != '\n' && t[e] != '\r') ++e;
        CountEmit em{counts};
        int rc = slow_line(v, t, pos, e, em);
        if (rc) { *exc = rc; *err_off = pos; return SVJG_E_INPUT; }
        ++*n_lines;
        if (e < n && t[e] == '\r' && e + 1 < n && t[e + 1] == '\n') ++e;
        pos = e + 1;
    }
    return 0;
}

// every node must be found by both lookups; probes around every node must agree between the two
extern "C" uint64_t hostsim_check_lookup(const svjg_graph *g) {
    std::vector<uint32_t> hash = build_chrom_hash(*g);
    BucketTable bt = build_buckets(*g);
    GraphView v = make_view(g, hash, bt);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < g->n_nodes; ++i) {
        uint64_t key = g->nodes[i].key;
        uint32_t c = (uint32_t)(key >> 48), pos = (uint32_t)(key >> 16);
        if (node_search(v, c, key) != i || node_lookup(v, c, pos, key) != i) ++bad;
        for (int d = -2; d <= 2; ++d) {
            uint64_t k2 = key + ((int64_t)d << 16);
            uint32_t p2 = (uint32_t)(k2 >> 16);
            if ((uint32_t)(k2 >> 48) != c) continue;
            if (node_search(v, c, k2) != node_lookup(v, c, p2, k2)) ++bad;
        }
    }
    return bad;
}
