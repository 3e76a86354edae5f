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

static GraphView make_view(const svjg_graph *g, const std::vector<uint32_t> &hash) {
    GraphView v;
    v.nodes = g->nodes; v.n_nodes = (uint32_t)g->n_nodes; v.edges = g->edges; v.hits = g->hits;
    v.chrom_names = (const uint8_t *)g->chrom_names; v.chrom_off = g->chrom_off; v.chrom_lo = g->chrom_node_lo;
    v.chrom_hash = hash.data(); v.n_chrom = g->n_chrom; v.hash_mask = (uint32_t)hash.size() - 1; v.d_over = g->d_over;
    v.name_tab = nullptr; v.name_mask = 0; v.link_tab = nullptr; v.link_mask = 0;     // main-kernel tables: not used by the exact path
    return v;
}

extern "C" int hostsim_classify(const svjg_graph *g, const char *gaf, uint64_t n,
                                uint32_t *counts, uint64_t *n_lines, int *exc, uint64_t *err_off)
{
    std::vector<uint32_t> hash = build_chrom_hash(*g);
    GraphView v = make_view(g, hash);
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

// the main kernel's hash tables must agree with the sorted node table / CSR rows: every node name resolves to its
// id, every CSR entry is found under its key with the same hits
extern "C" uint64_t hostsim_check_tables(const svjg_graph *g) {
    KernelTables kt = build_kernel_tables(*g);
    uint64_t bad = 0, found = 0;
    for (uint64_t j = 0; j <= kt.name_mask; ++j) {
        const uint32_t *e = &kt.names[j * NAME_ENT_WORDS];
        if (!(e[8] & 0xFFu)) continue;
        ++found;
        uint64_t q = name_hash_host(e, e[8] & 0xFFu) & kt.name_mask, steps = 0;
        while (q != j && steps <= kt.name_mask) { if (!(kt.names[q * NAME_ENT_WORDS + 8] & 0xFFu)) { ++bad; break; } q = (q + 1) & kt.name_mask; ++steps; }
        const svjg_node &nd = g->nodes[e[9]];
        uint32_t kind = (uint32_t)(nd.key >> 15) & 1u, pos = (uint32_t)(nd.key >> 16);
        if (e[10] != (kind ? nd.aux : nd.aux - pos + 1)) ++bad;
    }
    if (found > g->n_nodes) ++bad;
    for (uint64_t n = 0; n < g->n_nodes; ++n)
        for (uint32_t i = g->nodes[n].row & 0x7FFFFFFFu; i < (g->nodes[n + 1].row & 0x7FFFFFFFu); ++i) {
            const svjg_edge &ed = g->edges[i];
            uint64_t key = ((uint64_t)n << 33) | ((uint64_t)(ed.meta & 1u) << 32) | ((uint64_t)ed.right << 1) | ((ed.meta >> 1) & 1u);
            uint64_t q = link_hash_host(key) & kt.link_mask;
            for (;;) {
                const uint32_t *e = &kt.links[q * LINK_ENT_WORDS];
                if (e[0] == 0xFFFFFFFFu && e[1] == 0xFFFFFFFFu) { ++bad; break; }
                if (e[0] == (uint32_t)key && e[1] == (uint32_t)(key >> 32)) {
                    const uint32_t nh = ed.meta >> 2;
                    if (nh == 1 ? (e[2] != ed.h0 || e[3] != LINK_NO_HIT) : nh == 2 ? (e[2] != ed.h0 || e[3] != ed.h1) : (e[2] != (LINK_MANY | ed.h0) || e[3] != nh)) ++bad;
                    break;
                }
                q = (q + 1) & kt.link_mask;
            }
        }
    return bad;
}
