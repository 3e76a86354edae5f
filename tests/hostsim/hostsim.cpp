// TEST HARNESS ONLY (never shipped, never loaded by the product): compiles the exact per-line device routine
// (svjg::slow_line) and the graph-table lookups of svjedi-graph_amd/csrc/svjg_line.h with g++ and drives them
// sequentially, so their logic can be checked against the oracle on a machine without a GPU (and under
// -fsanitize=address,undefined).  The main kernel (k_classify_main) is wave/block-level code and is covered
// by the -m gpu tests through the C ABI.
#define SVJG_HD inline
#define SVJG_FN inline
#include "../../svjedi-graph_amd/csrc/svjg_line.h"
#include "../../svjedi-graph_amd/csrc/svjg_host_tables.h"
#include "../../svjedi-graph_amd/csrc/svjg_planes.h"
#include "../../svjedi-graph_amd/csrc/svjg_pass.h"
#include <cstring>
#include <string>
#include <vector>

using namespace svjg;

struct CountEmit {
    uint32_t *counts;
    void operator()(uint32_t slot, uint32_t allele) { counts[slot * 2 + allele]++; }
};

static GraphView make_view(const svjg_graph *g, const std::vector<uint32_t> &hash, const KernelTables *kt = nullptr) {
    GraphView v;
    v.nodes = g->nodes; v.n_nodes = (uint32_t)g->n_nodes; v.edges = g->edges; v.hits = g->hits;
    v.chrom_names = (const uint8_t *)g->chrom_names; v.chrom_off = g->chrom_off; v.chrom_lo = g->chrom_node_lo;
    v.chrom_hash = hash.data(); v.n_chrom = g->n_chrom; v.hash_mask = (uint32_t)hash.size() - 1; v.d_over = g->d_over;
    v.dover_list = (g->flags & SVJG_GRAPH_DOVER_LIST) ? 1u : 0u;
    v.node_of_kid = nullptr; v.name_tab = nullptr; v.name_pfx = nullptr; v.name_ihits = nullptr; v.name_disp = nullptr; v.name_slots = 0; v.name_buckets = 0; v.name_complete = 0; v.link_tab = nullptr; v.link_mask = 0; v.link_seed = 0;
    if (kt) {                                                             // the exact path resolves names through the node-name table
        v.node_of_kid = kt->node_of_kid.data(); v.name_tab = kt->names.data(); v.name_ihits = kt->ihits.data(); v.name_disp = kt->disp.data(); v.name_slots = kt->name_slots; v.name_buckets = kt->name_buckets;
        v.name_complete = (kt->names_left_out == 0 && kt->names_skipped == 0) ? 1u : 0u;
    }
    return v;
}

static int classify_view(const GraphView &v, const svjg_graph *g, const char *gaf, uint64_t n,
                         uint32_t *counts, uint64_t *n_lines, int *exc, uint64_t *err_off)
{
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
                // the table of the path's pieces, as k_classify_slow_wave builds it (there 64 bytes per step); offsets relative to the line
                // start like the kernel's staged copy (below 65536 for the lines the harness is given; longer ones: byte-by-byte search)
                std::vector<uint32_t> pieces; std::vector<uint16_t> colons; std::vector<uint64_t> keys;
                const uint8_t *tl = t + pos;
                SlowLine ll = ln; ll.ps -= pos; ll.pe -= pos;
                if (e - pos < 65536) {
                    const uint8_t s1 = ln.oriented ? '<' : ',', s2 = ln.oriented ? '>' : ',';
                    for (uint64_t q = ll.ps; q < ll.pe; ++q) {
                        const uint8_t c = tl[q], pc = q > ll.ps ? tl[q - 1] : s1;
                        if (c != s1 && c != s2 && (pc == s1 || pc == s2)) pieces.push_back((uint32_t)q);
                    }
                    for (size_t i = 0; i < pieces.size(); ++i) {
                        uint32_t e0 = i + 1 < pieces.size() ? pieces[i + 1] - 1u : (uint32_t)ll.pe;
                        while (e0 > pieces[i] && (tl[e0 - 1] == s1 || tl[e0 - 1] == s2)) --e0;
                        pieces[i] |= (e0 - pieces[i]) << 16;
                    }
                    if (!(g->flags & 64u))                                 // (flag 64, harness only: without the table of the pieces' colons)
                        for (size_t i = 0; i < pieces.size(); ++i) {
                            colons.push_back(piece_colons(tl, pieces[i] & 0xFFFFu, pieces[i] >> 16));
                            keys.push_back(piece_key(tl, pieces[i] & 0xFFFFu, colons.back()));
                        }
                    if (pieces.size() != ln.k) { *exc = 7; *err_off = pos; return SVJG_E_INPUT; }   // (harness self-check: pieces == nodes)
                }
                uint64_t best = ~0ull;
                if ((g->flags & 128u) && !pieces.empty()) {
                    // flag 128, harness only: phase 1 as k_classify_slow_wave runs it since r05 — every node resolved first, then the strands,
                    // by id where the line is clean (svjg_line.h: slow_wave_strands)
                    for (uint32_t lane = 0; lane < 64; ++lane) slow_wave_resolve(v, tl, ll, ns, lane, 64u, pieces.data());
                    bool clean = ll.oriented, rises = true;
                    for (uint32_t i = 0; i < ll.k; ++i) {
                        clean = clean && slow_node_clean(v, id[i]);
                        rises = rises && id[i] != NONE32 && (i == 0 || (id[1] > id[0] ? id[i] > id[i - 1] : id[i] < id[i - 1]));
                    }
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        uint64_t order = 0;
                        int r = slow_wave_strands(tl, ll, ns, lane, 64u, &order, pieces.data(), colons.empty() ? nullptr : colons.data(), keys.empty() ? nullptr : keys.data(), clean, rises);
                        if (r && ((order << 3) | (uint64_t)r) < best) best = (order << 3) | (uint64_t)r;
                    }
                } else
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    uint64_t order = 0;
                    int r = pieces.empty() ? slow_wave_phase1(v, t, ln, ns, lane, 64u, &order) : slow_wave_phase1(v, tl, ll, ns, lane, 64u, &order, pieces.data(), colons.empty() ? nullptr : colons.data(), keys.empty() ? nullptr : keys.data());
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
        } else if (g->flags & 32u) {                                       // harness only: one lane with its per-node results kept (k_classify_slow since r04), the lanes' entries interleaved
            SlowLine ln;
            rc = slow_prologue(t, pos, e, ln);
            if (!rc && ln.k >= 2) {
                const uint32_t S = 64, lane = (uint32_t)(*n_lines % 64);
                std::vector<uint32_t> id((size_t)ln.k * S); std::vector<int64_t> len((size_t)ln.k * S); std::vector<uint8_t> nrc((size_t)ln.k * S), strand((size_t)ln.k * S);
                NodeScratch ns{id.data() + lane, len.data() + lane, nrc.data() + lane, strand.data() + lane, ln.k, S};
                uint64_t order = 0;
                rc = slow_wave_phase1(v, t, ln, ns, 0u, 1u, &order);
                if (!rc) rc = slow_wave_phase2(v, ln, ns, em, 0u, 1u, &order);
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

extern "C" int hostsim_classify(const svjg_graph *g, const char *gaf, uint64_t n,
                                uint32_t *counts, uint64_t *n_lines, int *exc, uint64_t *err_off)
{
    std::vector<uint32_t> hash = build_chrom_hash(*g);
    KernelTables kt = build_kernel_tables(*g);
    GraphView v = make_view(g, hash, (g->flags & 2u) ? nullptr : &kt);      // flags bit 1 (harness only): sorted-table search instead
    return classify_view(v, g, gaf, n, counts, n_lines, exc, err_off);
}

// many small GAF fragments against one graph (tests/test_oracle_cross_fuzz.py): fragment i = gaf[offs[i], offs[i + 1]); its counts go to
// counts[i * n_slots * 2 ..] (zeroed by the caller), the exception class it dies with (0: none) to exc[i]
extern "C" void hostsim_classify_cases(const svjg_graph *g, const char *gaf, const uint64_t *offs, uint64_t n_cases, uint32_t *counts, int *exc)
{
    std::vector<uint32_t> hash = build_chrom_hash(*g);
    KernelTables kt = build_kernel_tables(*g);
    GraphView v = make_view(g, hash, (g->flags & 2u) ? nullptr : &kt);
    const uint64_t stride = (uint64_t)(g->n_slots ? g->n_slots : 1u) * 2u;
    for (uint64_t i = 0; i < n_cases; ++i) {
        // (a copy of its own: the routines read up to eight bytes at a time, never beyond the line's end — a heap block of exactly the
        //  fragment's size lets AddressSanitizer see it if they did)
        std::vector<char> frag(gaf + offs[i], gaf + offs[i + 1]);
        uint64_t nl = 0, eo = 0;
        exc[i] = 0;
        classify_view(v, g, frag.data(), frag.size(), counts + i * stride, &nl, &exc[i], &eo);
    }
}

// the main kernel's hash tables must agree with the sorted node table / CSR rows: every node name resolves to its
// id, every CSR entry is found under its key with the same hits
extern "C" uint64_t hostsim_check_tables(const svjg_graph *g) {
    KernelTables kt = build_kernel_tables(*g);
    uint64_t bad = kt.names_left_out + kt.links_left_out, found = 0;
    for (uint64_t j = 0; j < kt.name_slots; ++j) {
        const uint32_t *e = &kt.names[j * NAME_ENT_WORDS];
        if (name_ent_empty(e)) { if (e[8] != REC_NO_LINK || e[10] != REC_NO_LINK || e[12] != REC_NO_LINK || e[14] != REC_NO_LINK) ++bad; continue; }
        ++found;
        uint32_t d[NAME_WORDS];
        name_ent_words(e, d);
        uint64_t h = name_prehash(d, name_ent_len(e));
        if (name_ent_len(e) > 4 * NAME_WORDS) {                 // 49..64 bytes: the bytes in front of the last 48 are part of the hash (and wait in name_pfx)
            if (kt.name_pfx.empty()) { ++bad; continue; }
            h += name_pfx_hash(&kt.name_pfx[(size_t)name_ent_id(e) * NAME_PFX_WORDS]);
        }
        if (j != name_slot(h, kt.disp[name_bucket(h, kt.name_buckets)], kt.name_slots)) ++bad;   // the kernel's one probe lands here
        const uint32_t id = kt.node_of_kid[name_ent_id(e)];    // (the record holds the kernel's id: walk order)
        if (!kt.node_has[id] || kt.node_slot[id] != j || kt.node_pre[id] != h) ++bad;
        const svjg_node &nd = g->nodes[id];
        uint32_t kind = (uint32_t)(nd.key >> 15) & 1u, pos = (uint32_t)(nd.key >> 16);
        if ((e[7] & ~REC_ROW_INLINE) != (kind ? nd.aux : nd.aux - pos + 1) && !(e[6] & NAME_FLAG_NOLEN)) ++bad;
        {                                                       // the record's words are the window words of the node's canonical name
            const uint32_t c = (uint32_t)(nd.key >> 48), cnt = (uint32_t)nd.key & 0x7FFFu;
            std::string nm(g->chrom_names + g->chrom_off[c], g->chrom_off[c + 1] - g->chrom_off[c]);
            nm += ":" + std::to_string(pos) + (kind ? "." + std::to_string(cnt) : "-" + std::to_string(nd.aux));
            uint32_t w[NAME_WORDS];
            if (nm.size() > 4 * NAME_WORDS) {
                name_windows(nm.data() + (nm.size() - 4 * NAME_WORDS), 0, 4 * NAME_WORDS, w);
                uint32_t pw[NAME_PFX_WORDS];
                name_prefix_words(nm.data(), 0, (uint32_t)nm.size(), pw);
                if (memcmp(pw, &kt.name_pfx[(size_t)name_ent_id(e) * NAME_PFX_WORDS], sizeof pw) != 0) ++bad;
            } else
            name_windows(nm.data(), 0, (uint32_t)nm.size(), w);
            if (nm.size() != name_ent_len(e) || memcmp(w, d, sizeof w) != 0) ++bad;
        }
        // inline links = rows of this node, with the hits of the CSR row; REC_ROW_INLINE only if no row is missing
        const uint32_t ra = nd.row & 0x7FFFFFFFu, rb = g->nodes[id + 1].row & 0x7FFFFFFFu;
        uint32_t live = 0, inl = 0;
        for (uint32_t i = ra; i < rb; ++i) live += (g->edges[i].meta >> 2) != 0;
        for (uint32_t w = rec_first_link(e); w < 16; w += 2) {
            const uint32_t *l = e + w;
            if (l[0] == REC_NO_LINK) continue;
            ++inl;
            bool ok = false;
            for (uint32_t i = ra; i < rb; ++i) {
                const svjg_edge &ed = g->edges[i];
                if (((kt.kid[ed.right] << 2) | (ed.meta & 3u)) != l[0]) continue;
                const uint32_t nh = ed.meta >> 2;
                if (nh == 1) ok = l[1] == ed.h0;
                else if (l[1] & REC_MANY) {
                    const uint32_t *hp = &kt.ihits[l[1] & ~REC_MANY];
                    ok = hp[0] == nh;
                    for (uint32_t q = 0; q < nh && ok; ++q) ok = hp[1 + q] == (nh <= 2 ? (q ? ed.h1 : ed.h0) : g->hits[ed.h0 + q]);
                }
            }
            if (!ok) ++bad;
            for (uint32_t w2 = rec_first_link(e); w2 < w; w2 += 2) if (e[w2] == l[0]) ++bad;     // no key twice
        }
        if ((e[7] & REC_ROW_INLINE) && inl != live) ++bad;
    }
    if (found > g->n_nodes) ++bad;
    uint64_t n_links = 0;
    for (uint64_t n = 0; n < g->n_nodes; ++n)
        for (uint32_t i = g->nodes[n].row & 0x7FFFFFFFu; i < (g->nodes[n + 1].row & 0x7FFFFFFFu); ++i) {
            const svjg_edge &ed = g->edges[i];
            uint64_t key = ((uint64_t)kt.kid[n] << 33) | ((uint64_t)(ed.meta & 1u) << 32) | ((uint64_t)kt.kid[ed.right] << 1) | ((ed.meta >> 1) & 1u);
            uint32_t s1, s2;
            if (!kt.node_has[n] || !kt.node_has[ed.right]) continue;          // such lines take the exact path
            link_slots(link_prehash(kt.node_pre[n], ed.meta & 1u, kt.node_pre[ed.right], (ed.meta >> 1) & 1u), kt.link_seed, kt.link_mask, s1, s2);
            const uint32_t *e = &kt.links[(uint64_t)s1 * LINK_ENT_WORDS];
            if (!(e[0] == (uint32_t)key && e[1] == (uint32_t)(key >> 32))) e = &kt.links[(uint64_t)s2 * LINK_ENT_WORDS];
            if (!(e[0] == (uint32_t)key && e[1] == (uint32_t)(key >> 32))) {
                if (!(kt.names[(size_t)kt.node_slot[n] * NAME_ENT_WORDS + 6] & NAME_FLAG_HAZARD)) ++bad;        // unplaced link: its left node must be flagged for the exact path
                continue;
            }
            const uint32_t nh = ed.meta >> 2;
            if (nh == 1 ? (e[2] != ed.h0 || e[3] != LINK_NO_HIT) : nh == 2 ? (e[2] != ed.h0 || e[3] != ed.h1) : (e[2] != (LINK_MANY | ed.h0) || e[3] != nh)) ++bad;
            ++n_links;
        }
    uint64_t occupied = 0;
    for (uint64_t j = 0; j <= kt.link_mask; ++j) occupied += !(kt.links[j * LINK_ENT_WORDS] == 0xFFFFFFFFu && kt.links[j * LINK_ENT_WORDS + 1] == 0xFFFFFFFFu);
    if (occupied != n_links) ++bad;
    return bad;
}

// table statistics (harness only): names left out / skipped, links left out, slots, buckets
extern "C" void hostsim_table_stats(const svjg_graph *g, uint64_t *out) {
    KernelTables kt = build_kernel_tables(*g);
    out[0] = kt.names_left_out; out[1] = kt.names_skipped; out[2] = kt.links_left_out + (kt.links_unplaced << 32); out[3] = kt.name_slots; out[4] = kt.name_buckets;
    out[5] = kt.link_mask + 1ull; out[6] = kt.ihits.size();
    uint64_t mx = 0; for (uint16_t d : kt.disp) if (d > mx) mx = d;
    out[7] = mx;
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

// the fused pass's decisions (svjg_pass.h), as libsvjg_hip.so takes them
extern "C" uint64_t hostsim_pass_repeat_word(uint32_t overflow_bits) { return pass_repeat_word(overflow_bits); }
extern "C" int hostsim_pass_repeats(int has_comm, uint32_t own_overflow_bits, uint64_t guard_repeat_sum) { return pass_repeats(has_comm != 0, own_overflow_bits, guard_repeat_sum) ? 1 : 0; }
extern "C" int hostsim_pass_counts_overflowed(uint64_t a, uint64_t b) { return pass_counts_overflowed(a, b) ? 1 : 0; }
extern "C" uint32_t hostsim_guard_words(void) { return GUARD_WORDS; }
