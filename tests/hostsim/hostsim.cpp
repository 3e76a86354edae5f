// TEST HARNESS ONLY (never shipped, never loaded by the product): compiles the exact per-line device routine
// (svjg::slow_line) and the graph-table lookups of svjedi-graph_amd/csrc/svjg_line.h with g++ and drives them
// sequentially, so their logic can be checked against the oracle on a machine without a GPU (and under
// -fsanitize=address,undefined).  The main kernel (k_classify_main) is wave/block-level code and is covered
// by the -m gpu tests through the C ABI.
#define SVJG_HD inline
#include "../../svjedi-graph_amd/csrc/svjg_line.h"
#include "../../svjedi-graph_amd/csrc/svjg_host_tables.h"
#include "../../svjedi-graph_amd/csrc/svjg_planes.h"
#include <cstring>
#include <vector>

using namespace svjg;

struct CountEmit {
    uint32_t *counts;
    void operator()(uint32_t slot, uint32_t allele) { counts[slot * 2 + allele]++; }
};

static GraphView make_view(const svjg_graph *g, const std::vector<uint32_t> &hash) {
    GraphView v{};
    v.nodes = g->nodes; v.n_nodes = (uint32_t)g->n_nodes; v.edges = g->edges; v.hits = g->hits;
    v.chrom_names = (const uint8_t *)g->chrom_names; v.chrom_off = g->chrom_off; v.chrom_lo = g->chrom_node_lo;
    v.chrom_hash = hash.data(); v.n_chrom = g->n_chrom; v.hash_mask = (uint32_t)hash.size() - 1; v.d_over = g->d_over;
    return v;                                                              // (the exact path does not touch the main kernel's tables)
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
        int rc = 0;
        if (g->flags & 8u) {                                               // harness only: the two-phase wave routine of k_classify_slow_wave, 64 lanes
            SlowLine ln;
            rc = slow_prologue(t, pos, e, ln);
            if (!rc && ln.k >= 2) {
                std::vector<uint32_t> id(ln.k); std::vector<int64_t> len(ln.k); std::vector<uint8_t> nrc(ln.k), strand(ln.k);
                NodeScratch ns{id.data(), len.data(), nrc.data(), strand.data(), ln.k};
                uint64_t best = ~0ull;
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    uint64_t order = 0;
                    int r = slow_wave_phase1(v, t, ln, ns, lane, 64u, &order);
                    if (r && ((order << 3) | (uint64_t)r) < best) best = (order << 3) | (uint64_t)r;
                }
                if (best == ~0ull)
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        uint64_t order = 0;
                        int r = slow_wave_phase2(v, ln, ns, em, lane, 64u, &order);
                        if (r && ((order << 3) | (uint64_t)r) < best) best = (order << 3) | (uint64_t)r;
                    }
                if (best != ~0ull) rc = (int)(best & 7);
            }
        } else if (g->flags & 4u) {                                        // harness only: 64 cooperating lanes, as k_classify_slow_wave runs a line with too many nodes
            uint64_t best = ~0ull;
            for (uint32_t lane = 0; lane < 64; ++lane) {
                uint64_t order = 0;
                int r = slow_line(v, t, pos, e, em, lane, 64u, &order);
                if (r && ((order << 3) | (uint64_t)r) < best) best = (order << 3) | (uint64_t)r;
            }
            if (best != ~0ull) rc = (int)(best & 7);
        } else rc = slow_line(v, t, pos, e, em);
        if (rc) { *exc = rc; *err_off = pos; return SVJG_E_INPUT; }
        ++*n_lines;
        if (e < n && t[e] == '\r' && e + 1 < n && t[e + 1] == '\n') ++e;
        pos = e + 1;
    }
    return 0;
}

// the main kernel's tables must agree with the sorted node table / CSR rows: every directed link of the graph is found — through
// the canonical form and the kernel's bucket probe(s) — with the same hits, every entry is some link's, and every chromosome name
// resolves to its index
extern "C" uint64_t hostsim_check_tables(const svjg_graph *g) {
    KernelTables kt = build_kernel_tables(*g);
    uint64_t bad = kt.links_left_out, n_found = 0;
    for (uint64_t n = 0; n < g->n_nodes; ++n) {
        uint32_t cl, al, bl, kl;
        node_fields(g->nodes[n], cl, al, bl, kl);
        for (uint32_t i = g->nodes[n].row & 0x7FFFFFFFu; i < (g->nodes[n + 1].row & 0x7FFFFFFFu); ++i) {
            const svjg_edge &ed = g->edges[i];
            const uint32_t nh = ed.meta >> 2;
            uint32_t cr, ar, br, kr;
            node_fields(g->nodes[ed.right], cr, ar, br, kr);
            bool flipped;
            const LinkKey k = link_key(al, bl, cl | (kl ? TAG_KIND : 0u) | ((ed.meta & 1u) ? TAG_STRAND : 0u), ar, br, cr | (kr ? TAG_KIND : 0u) | ((ed.meta & 2u) ? TAG_STRAND : 0u), flipped);
            const uint32_t *e = link_find(kt, k);
            if (!nh) { if (e) ++bad; continue; }
            if (!e) { ++bad; continue; }
            ++n_found;
            // same multiset of hits as the row (the reversed form's row lists them in the other order)
            std::vector<uint32_t> want, got;
            for (uint32_t q = 0; q < nh; ++q) want.push_back(nh <= 2 ? (q ? ed.h1 : ed.h0) : g->hits[ed.h0 + q]);
            if ((e[6] & LINK_MANY) && e[7] != LINK_NO_HIT) for (uint32_t q = 0; q < e[7]; ++q) got.push_back(g->hits[(e[6] & ~LINK_MANY) + q]);
            else { got.push_back(e[6]); if (e[7] != LINK_NO_HIT) got.push_back(e[7]); }
            std::sort(want.begin(), want.end()); std::sort(got.begin(), got.end());
            if (want != got) ++bad;
            // the alt node's length, or the exact-path flag
            const uint32_t k0 = flipped ? kr : kl, k1 = flipped ? kl : kr;
            if ((((e[4] >> 14) & 1u) | (((e[4] >> 30) & 1u) << 1)) != (k0 | (k1 << 1))) ++bad;
            const uint32_t alt = e[5] >> LK_ALT_SHIFT;
            if (kl && kr) { if (!(e[5] & LKF_EXACT)) ++bad; }
            else if (kl || kr) {
                const uint32_t len = kl ? g->nodes[n].aux : g->nodes[ed.right].aux;
                if (len == SVJG_LEN_UNKNOWN || len >= (1u << 25)) { if (!(e[5] & LKF_EXACT)) ++bad; }
                else if (alt != len || (e[5] & LKF_EXACT)) ++bad;
            } else if (alt != LK_ALT_NONE || (e[5] & LKF_EXACT)) ++bad;
        }
    }
    uint64_t occupied = 0;
    for (uint64_t j = 0; j < (uint64_t)kt.l_buckets * 2; ++j) occupied += kt.links[j * LK_WORDS + 4] != 0xFFFFFFFFu;
    if (occupied != kt.n_keys || n_found < kt.n_keys || n_found > 2 * kt.n_keys) ++bad;   // (every entry answers one or two directed rows)
    for (uint32_t c = 0; c < g->n_chrom; ++c) {
        const uint32_t o = g->chrom_off[c], n = g->chrom_off[c + 1] - o;
        if (n == 0 || n > 24) continue;
        uint32_t w[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t b = 0; b < n; ++b) w[b >> 2] |= (uint32_t)(uint8_t)g->chrom_names[o + b] << (8 * (b & 3));
        uint32_t meta = CT_EMPTY;
        if (n <= 8) {
            for (uint32_t j = chrom_short_slot(w[0], w[1], kt.cs_mask, kt.cs_mult), probes = 0;; j = (j + 1) & kt.cs_mask, ++probes) {
                if (probes && !(kt.cs_mult & 1u)) { ++bad; break; }         // (one probe must decide)
                const uint32_t *e = &kt.cshort[(size_t)j * 4];
                if (e[2] == CT_EMPTY) break;
                if (e[0] == w[0] && e[1] == w[1] && (e[2] >> 24) == n) { meta = e[2]; break; }
            }
        } else {
            for (uint32_t j = chrom_long_slot(w, kt.cl_mask);; j = (j + 1) & kt.cl_mask) {
                const uint32_t *e = &kt.clong[(size_t)j * 8];
                if (e[6] == CT_EMPTY) break;
                bool eq = (e[6] >> 24) == n;
                for (int q = 0; q < 6; ++q) eq = eq && e[q] == w[q];
                if (eq) { meta = e[6]; break; }
            }
        }
        if (meta == CT_EMPTY || (meta & 0xFFFFu) != c) ++bad;
    }
    return bad;
}

// table statistics (harness only): canonical links, links in their second bucket, links left out, buckets, links flagged for the
// exact path, chromosome names too long for the tables, entries of the two chromosome tables
extern "C" void hostsim_table_stats(const svjg_graph *g, uint64_t *out) {
    KernelTables kt = build_kernel_tables(*g);
    out[0] = kt.n_keys; out[1] = kt.n_second; out[2] = kt.links_left_out; out[3] = kt.l_buckets; out[4] = kt.n_exact;
    out[5] = kt.chroms_skipped; out[6] = kt.cs_mask + 1ull; out[7] = kt.cl_mask + 1ull;
}

// byte classes of n 64-byte spans by bit planes (svjg_planes.h, phase B1 of the main kernel): out = 8 masks of 64 bits per span
// (nl, cr, tab, ori, nd, dee, colon, high)
extern "C" void hostsim_span_classes(const uint8_t *text, uint64_t n_spans, uint64_t *out) {
    for (uint64_t s = 0; s < n_spans; ++s) {
        uint32_t w[16];
        memcpy(w, text + s * 64, 64);
        span_planes(w);
        const HalfClasses lo = half_classes(w), hi = half_classes(w + 8);
        uint64_t *o = out + s * 8;
        o[0] = lo.nl | ((uint64_t)hi.nl << 32); o[1] = lo.cr | ((uint64_t)hi.cr << 32); o[2] = lo.tab | ((uint64_t)hi.tab << 32);
        o[3] = lo.ori | ((uint64_t)hi.ori << 32); o[4] = lo.nd | ((uint64_t)hi.nd << 32); o[5] = lo.dee | ((uint64_t)hi.dee << 32);
        o[6] = lo.colon | ((uint64_t)hi.colon << 32); o[7] = lo.high | ((uint64_t)hi.high << 32);
    }
}
