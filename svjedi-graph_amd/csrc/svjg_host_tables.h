// Host-side table builders shared by the library's graph upload and the host test harness:
// the open-addressing chromosome-name hash table that svjg::chrom_lookup probes (exact path) and the
// name / link hash tables of the main kernel.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
#include "../../include/svjg.h"

namespace svjg {

inline uint32_t fnv1a32_host(const char *s, uint32_t n) {
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < n; ++i) h = (h ^ (uint8_t)s[i]) * 16777619u;
    return h;
}

// value = chrom index + 1, 0 = empty; size is a power of two >= 2 * n_chrom + 2
inline std::vector<uint32_t> build_chrom_hash(const svjg_graph &g) {
    uint32_t sz = 8;
    while (sz < 2 * g.n_chrom + 2) sz *= 2;
    std::vector<uint32_t> tab(sz, 0);
    for (uint32_t c = 0; c < g.n_chrom; ++c) {
        uint32_t o = g.chrom_off[c], n = g.chrom_off[c + 1] - o;
        uint32_t j = fnv1a32_host(g.chrom_names + o, n) & (sz - 1);
        while (tab[j]) j = (j + 1) & (sz - 1);
        tab[j] = c + 1;
    }
    return tab;
}

}  // namespace svjg

#include <algorithm>
#include <string>
#include <unordered_map>
#include "svjg_line.h"

namespace svjg {

// ---- tables of the main kernel ---------------------------------------------------------------------------------
// The kernel parses a path node "chrom:start-end" / "chrom:pos.n" into numbers and asks ONE table, keyed by the link
// (svjg_line.h: link_key / link_hash / link_bucket1): one 64-byte bucket per path step, no node lookup (the reference has
// none either: filter-alignments.py:141-153, :343-349).  Random 64-byte lines of a table of tens of MB come at ~66 G lines/s
// on MI355X and at ~110 G/s from 8 MB (tools/ubench/randread.hip), so the table is small: canonical links only (a link
// and its reversed form share an entry), 32 bytes each.
//   CHROMOSOME TABLES: raw name bytes -> index, linear probing at load <= 1/4: names of up to 8 bytes (two words to
//   compare; every usual name) and names of 9..24 bytes.  Longer names stay out: their lines take the exact path.
struct KernelTables {
    std::vector<uint32_t> links; uint32_t l_buckets = 0;      // LB_WORDS words per bucket
    std::vector<uint32_t> cshort, clong; uint32_t cs_mask = 0, cl_mask = 0, cs_mult = 0x9E3779B1u;
    uint64_t n_keys = 0, n_second = 0, n_exact = 0;           // canonical links, those in their second bucket, those flagged for the exact path
    uint64_t links_left_out = 0;                              // links no bucket could take (never seen: the library then uses the exact path only)
    uint32_t chroms_skipped = 0;                              // names longer than 24 bytes
};

struct CanonLink { LinkKey k; uint32_t f5, h0, h1; uint64_t h; };
constexpr uint32_t MAIN_MAX_CHROM = TAG_CHROM;              // chromosome indices the main kernel's node tags hold

#ifndef SVJG_LINK_LOAD
#define SVJG_LINK_LOAD 0.75
#endif

inline void build_chrom_tables(const svjg_graph &g, KernelTables &kt) {
    uint32_t n_s = 0, n_l = 0;
    for (uint32_t c = 0; c < g.n_chrom; ++c) { const uint32_t n = g.chrom_off[c + 1] - g.chrom_off[c]; if (n >= 1 && n <= 8) ++n_s; else if (n <= 24) ++n_l; }
    uint32_t ss = 64, sl = 16;
    while (ss < 4 * n_s) ss *= 2;
    while (sl < 4 * n_l) sl *= 2;
    kt.cshort.assign((size_t)ss * 4, CT_EMPTY); kt.cs_mask = ss - 1;
    // a multiplier under which no two short names share a slot: the kernel then needs no probing loop (cs_mult's bit 0 clear says so)
    {
        std::vector<std::pair<uint32_t, uint32_t>> names;
        for (uint32_t c = 0; c < g.n_chrom; ++c) {
            const uint32_t o = g.chrom_off[c], n = g.chrom_off[c + 1] - o;
            if (n < 1 || n > 8) continue;
            uint32_t w[2] = {0, 0};
            for (uint32_t b = 0; b < n; ++b) w[b >> 2] |= (uint32_t)(uint8_t)g.chrom_names[o + b] << (8 * (b & 3));
            names.emplace_back(w[0], w[1]);
        }
        std::vector<uint8_t> used(ss);
        uint32_t m = 0x9E3779B1u;
        bool found = false;
        for (int attempt = 0; attempt < 4096 && !found; ++attempt, m = m * 0x2C1B3C6Du + 0x7F4A7C15u) {
            std::fill(used.begin(), used.end(), 0);
            found = true;
            for (const auto &nm : names) {
                const uint32_t j = chrom_short_slot(nm.first, nm.second, kt.cs_mask, m);
                if (used[j]) { found = false; break; }
                used[j] = 1;
            }
            if (found) kt.cs_mult = m & ~1u;
        }
        if (!found) kt.cs_mult = 0x9E3779B1u;                  // (bit 0 set: probing loop)
    }
    kt.clong.assign((size_t)sl * 8, CT_EMPTY); kt.cl_mask = sl - 1;
    // classes of names of which one ends with another ("1" / "11"): only nodes of one class can be substrings of one another
    // (SURVEY Q6); the kernel compares start | kind + class * constant when it looks for a line's repeated names
    std::vector<uint32_t> cls(g.n_chrom);
    for (uint32_t c = 0; c < g.n_chrom; ++c) cls[c] = c;
    auto root = [&](uint32_t x) { while (cls[x] != x) x = cls[x] = cls[cls[x]]; return x; };
    {
        std::unordered_map<std::string, uint32_t> by_name;
        for (uint32_t c = 0; c < g.n_chrom; ++c) by_name.emplace(std::string(g.chrom_names + g.chrom_off[c], g.chrom_off[c + 1] - g.chrom_off[c]), c);
        for (uint32_t d = 0; d < g.n_chrom; ++d) {
            const std::string nm(g.chrom_names + g.chrom_off[d], g.chrom_off[d + 1] - g.chrom_off[d]);
            for (size_t cut = 1; cut < nm.size(); ++cut) {               // every proper suffix of the name that is a name itself
                const auto it = by_name.find(nm.substr(cut));
                if (it == by_name.end()) continue;
                const uint32_t a = root(it->second), b = root(d);
                if (a != b) cls[a > b ? a : b] = a > b ? b : a;
            }
        }
    }
    for (uint32_t c = 0; c < g.n_chrom; ++c) {
        const uint32_t o = g.chrom_off[c], n = g.chrom_off[c + 1] - o;
        if (n == 0 || n > 24) { ++kt.chroms_skipped; continue; }
        uint32_t w[6] = {0, 0, 0, 0, 0, 0}, flags = 0;
        for (uint32_t b = 0; b < n; ++b) { w[b >> 2] |= (uint32_t)(uint8_t)g.chrom_names[o + b] << (8 * (b & 3)); if (g.chrom_names[o + b] == ':') flags |= CT_ODD; }
        const uint32_t meta = c | (flags << 16) | (n << 24), hc = root(c);
        if (n <= 8) {
            uint32_t j = chrom_short_slot(w[0], w[1], kt.cs_mask, kt.cs_mult);
            while (kt.cshort[(size_t)j * 4 + 2] != CT_EMPTY) j = (j + 1) & kt.cs_mask;
            kt.cshort[(size_t)j * 4] = w[0]; kt.cshort[(size_t)j * 4 + 1] = w[1]; kt.cshort[(size_t)j * 4 + 2] = meta; kt.cshort[(size_t)j * 4 + 3] = hc;
        } else {
            uint32_t j = chrom_long_slot(w, kt.cl_mask);
            while (kt.clong[(size_t)j * 8 + 6] != CT_EMPTY) j = (j + 1) & kt.cl_mask;
            for (int q = 0; q < 6; ++q) kt.clong[(size_t)j * 8 + q] = w[q];
            kt.clong[(size_t)j * 8 + 6] = meta; kt.clong[(size_t)j * 8 + 7] = hc;
        }
    }
}

inline void node_fields(const svjg_node &nd, uint32_t &c, uint32_t &a, uint32_t &b, uint32_t &k) {
    c = (uint32_t)(nd.key >> 48); a = (uint32_t)(nd.key >> 16); k = (uint32_t)(nd.key >> 15) & 1u;
    b = k ? (uint32_t)nd.key & 0x7FFFu : nd.aux;
}

inline KernelTables build_kernel_tables(const svjg_graph &g) {
    KernelTables kt;
    const bool verbose = getenv("SVJG_VERBOSE") != nullptr;          // stage timers on stderr (measurement only)
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[svjg] kernel tables, %s: %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    build_chrom_tables(g, kt);
    // lookup forms of the links: the CSR rows hold d[K] ++ d[R] under K and d[R] ++ d[K] under R (the same multiset: the order of the
    // appends of one line to different lists does not show in the output)
    std::vector<CanonLink> keys;
    keys.reserve((size_t)g.n_edges / 2 + 16);
    for (uint64_t n = 0; n < g.n_nodes; ++n) {
        const uint32_t ra = g.nodes[n].row & 0x7FFFFFFFu, rb = g.nodes[n + 1].row & 0x7FFFFFFFu;
        if (ra == rb) continue;
        uint32_t cl, al, bl, kl;
        node_fields(g.nodes[n], cl, al, bl, kl);
        for (uint32_t i = ra; i < rb; ++i) {
            const svjg_edge &ed = g.edges[i];
            const uint32_t nh = ed.meta >> 2;
            if (!nh) continue;
            uint32_t cr, ar, br, kr;
            node_fields(g.nodes[ed.right], cr, ar, br, kr);
            bool flipped;
            CanonLink cl_;
            const uint32_t sl = ed.meta & 1u, sr = (ed.meta >> 1) & 1u;
            cl_.k = link_key(al, bl, cl | (kl ? TAG_KIND : 0u) | (sl ? TAG_STRAND : 0u), ar, br, cr | (kr ? TAG_KIND : 0u) | (sr ? TAG_STRAND : 0u), flipped);
            // lookup form = this row's own form (left strand '+'), or — a (-,+) row — its reversed form, which is another (-,+) row's
            // own spelling; the reversed form of a (-,-) row is a (+,+) row, entered there
            if (flipped && sr) continue;
            // the alt node's length travels with the link (the kernel has no node table): unknown / too long / two alt nodes -> exact path
            uint32_t alt = LK_ALT_NONE, f5 = 0;
            if (kl && kr) f5 |= LKF_EXACT;
            else if (kl || kr) {   // (which side the alt node is on in the entry: the tags' kind bits)
                const uint32_t len = kl ? g.nodes[n].aux : g.nodes[ed.right].aux;
                if (len == SVJG_LEN_UNKNOWN || len >= (1u << 25)) f5 |= LKF_EXACT; else alt = len;
            }
            cl_.f5 = f5 | (alt << LK_ALT_SHIFT);
            if (nh == 1) { cl_.h0 = ed.h0; cl_.h1 = LINK_NO_HIT; }
            else if (nh == 2) { cl_.h0 = ed.h0; cl_.h1 = ed.h1; }
            else { cl_.h0 = LINK_MANY | ed.h0; cl_.h1 = nh; }
            cl_.h = link_hash(cl_.k);
            if (f5 & LKF_EXACT) ++kt.n_exact;
            keys.push_back(cl_);
        }
    }
    kt.n_keys = keys.size();
    lap("lookup forms of the links");
    const char *ld = getenv("SVJG_LINK_LOAD");                       // measurement only: keys per bucket
    double load = ld ? atof(ld) : SVJG_LINK_LOAD;
    if (!(load > 0.05 && load <= 1.5)) load = SVJG_LINK_LOAD;
    uint64_t nb = (uint64_t)((double)keys.size() / load) + 16;
    for (int grow = 0;; ++grow) {
        // every key goes to its first bucket while that has room; the others to their second bucket, evicting (random walk between
        // the two buckets of the evicted key) when that is full too: at <= 1 key per two-entry bucket a handful of moves at most
        kt.l_buckets = (uint32_t)nb;
        std::vector<int64_t> owner((size_t)nb * 2, -1);
        std::vector<uint32_t> later;
        auto b1_of = [&](int64_t i) { return link_bucket1(keys[(size_t)i].h, kt.l_buckets); };
        auto b2_of = [&](int64_t i) { return link_bucket2(keys[(size_t)i].h, kt.l_buckets, b1_of(i)); };
        auto try_put = [&](uint32_t b, int64_t i) { for (int q = 0; q < 2; ++q) if (owner[(size_t)b * 2 + q] < 0) { owner[(size_t)b * 2 + q] = i; return true; } return false; };
        for (size_t i = 0; i < keys.size(); ++i) if (!try_put(b1_of((int64_t)i), (int64_t)i)) later.push_back((uint32_t)i);
        uint64_t left = 0, rng = 0x9E3779B97F4A7C15ull;
        for (uint32_t i0 : later) {
            int64_t cur = (int64_t)i0;
            uint32_t at = b1_of(cur);                              // the bucket `cur` has just failed in / been evicted from
            bool placed = false;
            for (int kick = 0; kick < 256 && !placed; ++kick) {
                const uint32_t b1 = b1_of(cur), b2 = b2_of(cur), to = at == b1 ? b2 : b1;
                if (try_put(to, cur)) { placed = true; break; }
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                int64_t &victim = owner[(size_t)to * 2 + ((rng >> 33) & 1u)];
                std::swap(cur, victim);
                at = to;
            }
            if (!placed) ++left;
        }
        if (left && grow < 8) { nb += nb / 2; continue; }
        kt.links_left_out = left + (g.n_chrom > MAIN_MAX_CHROM ? 1u : 0u);   // (more chromosomes than a node tag holds: exact path only)
        kt.links.assign((size_t)nb * LB_WORDS, 0xFFFFFFFFu);
        kt.n_second = 0;
        for (size_t sl = 0; sl < owner.size(); ++sl) {
            if (owner[sl] < 0) continue;
            const CanonLink &c = keys[(size_t)owner[sl]];
            uint32_t *e = &kt.links[sl * LK_WORDS];
            e[0] = c.k.w0; e[1] = c.k.w1; e[2] = c.k.w2; e[3] = c.k.w3; e[4] = c.k.w4; e[5] = c.f5; e[6] = c.h0; e[7] = c.h1;
        }
        for (size_t sl = 0; sl < owner.size(); ++sl) {
            if (owner[sl] < 0) continue;
            const uint32_t b1 = b1_of(owner[sl]);
            if (sl / 2 != b1) { kt.links[(size_t)b1 * LB_WORDS + 5] |= LKF_OVER; ++kt.n_second; }   // (a key is evicted from full buckets only: entry 0 of its first bucket is in use)
        }
        break;
    }
    lap("link table placement");
    return kt;
}

// the kernel's lookup, on the host (table checks, tests/hostsim): entry of the directed link or nullptr; flipped as link_key says
inline const uint32_t *link_find(const KernelTables &kt, const LinkKey &k) {
    const uint64_t h = link_hash(k);
    const uint32_t b1 = link_bucket1(h, kt.l_buckets);
    const uint32_t *e = &kt.links[(size_t)b1 * LB_WORDS];
    if (e[4] != 0xFFFFFFFFu && link_match(e, k)) return e;
    if (e[LK_WORDS + 4] != 0xFFFFFFFFu && link_match(e + LK_WORDS, k)) return e + LK_WORDS;
    if (e[4] == 0xFFFFFFFFu || !(e[5] & LKF_OVER)) return nullptr;
    const uint32_t b2 = link_bucket2(h, kt.l_buckets, b1);
    e = &kt.links[(size_t)b2 * LB_WORDS];
    if (e[4] != 0xFFFFFFFFu && link_match(e, k)) return e;
    if (e[LK_WORDS + 4] != 0xFFFFFFFFu && link_match(e + LK_WORDS, k)) return e + LK_WORDS;
    return nullptr;
}

}  // namespace svjg
