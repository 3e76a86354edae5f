// TEST HARNESS ONLY (never shipped, never loaded by the product): compiles the per-line device routines of
// svjedi-graph_amd/csrc/svjg_line.h with g++ and drives them sequentially, so their logic can be checked
// against the oracle on a machine without a GPU (and under -fsanitize=address,undefined).
// The wave-level parts of the kernels (staging, newline scan, commit, atomics) are NOT exercised here;
// those are covered by the -m gpu tests through the C ABI.
#define SVJG_HD inline
#include "../../svjedi-graph_amd/csrc/svjg_line.h"
#include "../../svjedi-graph_amd/csrc/svjg_host_tables.h"
#include <vector>

using namespace svjg;

struct CountEmit {
    uint32_t *counts;
    void operator()(uint32_t slot, uint32_t allele) { counts[slot * 2 + allele]++; }
};

static uint64_t g_reasons[32];
extern "C" void hostsim_defer_reasons(uint64_t *out) { for (int i = 0; i < 32; ++i) { out[i] = g_reasons[i]; g_reasons[i] = 0; } }

extern "C" int hostsim_classify(const svjg_graph *g, const char *gaf, uint64_t n, int force_slow, uint32_t pend_cap,
                                uint32_t *counts, uint64_t *n_lines, uint64_t *n_deferred, int *exc, uint64_t *err_off)
{
    std::vector<uint32_t> hash = build_chrom_hash(*g);
    GraphView v;
    v.nodes = g->nodes; v.n_nodes = (uint32_t)g->n_nodes; v.edges = g->edges; v.hits = g->hits;
    v.chrom_names = (const uint8_t *)g->chrom_names; v.chrom_off = g->chrom_off; v.chrom_lo = g->chrom_node_lo;
    v.chrom_hash = hash.data(); v.n_chrom = g->n_chrom; v.hash_mask = (uint32_t)hash.size() - 1; v.d_over = g->d_over;
    const uint8_t *t = (const uint8_t *)gaf;
    std::vector<Pending> pend(pend_cap ? pend_cap : 1);
    *n_lines = 0; *n_deferred = 0; *exc = 0; *err_off = 0;
    uint64_t pos = 0;
    while (pos < n) {
        uint64_t e = pos;
        while (e < n && t[e] != '\n' && t[e] != '\r') ++e;
        uint32_t m = 0;
        int st = (force_slow || (g->flags & SVJG_GRAPH_ALL_SLOW) || e - pos > 60000) ? -31
                 : fast_line(v, t + pos, 0u, (uint32_t)(e - pos), pend.data(), pend_cap, &m);
        if (st >= 0) {
            for (uint32_t i = 0; i < m; ++i) {
                counts[pend[i].hit * 2] += pend[i].pre & 0xFFFF;
                counts[pend[i].hit * 2 + 1] += pend[i].pre >> 16;
            }
        } else {
            ++*n_deferred;
            g_reasons[st < 0 && st > -32 ? -st : 0]++;
            CountEmit em{counts};
            int rc = slow_line(v, t, pos, e, em);
            if (rc) { *exc = rc; *err_off = pos; return SVJG_E_INPUT; }
        }
        ++*n_lines;
        if (e < n && t[e] == '\r' && e + 1 < n && t[e + 1] == '\n') ++e;
        pos = e + 1;
    }
    return 0;
}
