// Host-side table builders shared by the library's graph upload and the host test harness:
// the open-addressing chromosome-name hash table that svjg::chrom_lookup probes (exact path) and the
// name / link hash tables of the main kernel.
#pragma once
#include <stddef.h>
#include <stdint.h>
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

#include <string>

namespace svjg {

// ---- hash tables of the main kernel --------------------------------------------------------------------
// Both tables are two-choice (cuckoo) tables built here on the host: every key sits in one of its TWO candidate
// slots, so the kernel fetches both candidates at once and a lookup is exactly one round trip to memory for every
// lane of a wave (with linear probing, some lane of nearly every wave had to walk on: a second, dependent trip).
// NAME TABLE: canonical node name (<= 32 bytes, zero padded to eight words) -> node.  The kernel hashes the raw
// bytes of a path segment and compares them with the stored spelling: no number parsing on the device, and
// only names spelled exactly like the graph's can match (anything else goes to the exact path).
//   entry = 16 words (one 64-byte line): [0..5] name bytes 0..23, [6] node id << 7 | flags << 5 | (byte length - 1)
//                     (flags: bit 0 hazard-prone, bit 1 length unknown; all ones = empty slot), [7] node length in bp,
//                     [8..9] name bytes 24..31, [10..15] unused.  Names of up to 24 bytes (all of the usual
//                     "chrN:start-end") are decided by the first two 16-byte loads.
//   (A 16-byte fingerprint entry was measured too: 2 % faster, not worth giving up the exact comparison.)
// LINK TABLE: (left id, left strand, right id, right strand) -> hits, same content as the CSR rows.
//   entry = 4 words (one 16-byte load): [0] key low, [1] key high, [2] a, [3] b with
//       1 hit : a = hit, b = LINK_NO_HIT        2 hits: a, b = the hits
//       more  : a = LINK_MANY | index into hits[], b = number of hits
//   key = left << 33 | left strand << 32 | right << 1 | right strand ; all ones = empty slot.
// A key the builder cannot place (three keys with one pre-hash; never seen) is left out: the kernel then finds
// nothing and hands the line to the exact path.
constexpr uint32_t NAME_ENT_WORDS = 16, LINK_ENT_WORDS = 4;
constexpr uint32_t LINK_NO_HIT = 0xFFFFFFFFu, LINK_MANY = 0x80000000u;
constexpr uint32_t NAME_EMPTY = 0xFFFFFFFFu, NAME_MAX_ID = (1u << 25) - 2u;
inline bool name_ent_empty(const uint32_t *e) { return e[6] == NAME_EMPTY; }
inline uint32_t name_ent_len(const uint32_t *e) { return (e[6] & 31u) + 1u; }
inline uint32_t name_ent_id(const uint32_t *e) { return e[6] >> 7; }
inline void name_ent_words(const uint32_t *e, uint32_t d[8]) { for (int w = 0; w < 6; ++w) d[w] = e[w]; d[6] = e[8]; d[7] = e[9]; }

// pre-hash of a name (the kernel's name_words computes the same sum) and of a link key
inline uint32_t name_prehash_host(const uint32_t *d, uint32_t len) {
    static const uint32_t C[8] = {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u};
    uint32_t h = len * 0x7FEB352Du;
    for (int i = 0; i < 8; ++i) h += d[i] * C[i];
    return h;
}
inline uint32_t link_prehash_host(uint64_t key) { return (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u); }
// the two candidate slots of a pre-hash (svjg_kernels.h: cuckoo_slots is the device twin)
inline void cuckoo_slots_host(uint32_t x, uint32_t seed, uint32_t mask, uint32_t &s1, uint32_t &s2) {
    uint32_t a = x ^ seed;
    a ^= a >> 15; a *= 0x2C1B3C6Du; a ^= a >> 12;
    uint32_t b = (x + seed) * 0x85EBCA6Bu;
    b ^= b >> 13; b *= 0xC2B2AE35u; b ^= b >> 16;
    s1 = a & mask; s2 = b & mask;
    if (s2 == s1) s2 = s1 ^ 1u;
}

struct KernelTables {
    std::vector<uint32_t> names; uint32_t name_mask = 0, name_seed = 0;
    std::vector<uint32_t> links; uint32_t link_mask = 0, link_seed = 0;
    uint64_t names_left_out = 0, links_left_out = 0;
    uint32_t names_skipped = 0;                               // node names the table cannot hold (> 32 bytes, id too large)
};

// Two-choice placement by random-walk eviction.  pre[i] = pre-hash of key i; returns owner[slot] = key index or -1.
// Keys that cannot be placed under any of a few seeds are dropped (counted in left_out).
inline std::vector<int64_t> cuckoo_place(const std::vector<uint32_t> &pre, uint32_t mask, uint32_t &seed_out, uint64_t &left_out) {
    std::vector<int64_t> best;
    uint64_t best_left = ~0ull;
    for (uint32_t seed = 0x5bd1e995u, attempt = 0; attempt < 8; ++attempt, seed = seed * 0x9E3779B1u + 0x7F4A7C15u) {
        std::vector<int64_t> owner((size_t)mask + 1, -1);
        uint64_t left = 0, rng = 0x9E3779B97F4A7C15ull ^ seed;
        for (size_t i = 0; i < pre.size(); ++i) {
            int64_t cur = (int64_t)i;
            uint32_t avoid = 0xFFFFFFFFu;
            bool placed = false;
            for (int kick = 0; kick < 512; ++kick) {
                uint32_t s1, s2;
                cuckoo_slots_host(pre[(size_t)cur], seed, mask, s1, s2);
                if (owner[s1] < 0) { owner[s1] = cur; placed = true; break; }
                if (owner[s2] < 0) { owner[s2] = cur; placed = true; break; }
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                uint32_t victim = (rng >> 33) & 1u ? s1 : s2;
                if (victim == avoid) victim = victim == s1 ? s2 : s1;
                std::swap(cur, owner[victim]);
                avoid = victim;
            }
            if (!placed) ++left;                              // `cur` (whoever was evicted last) stays out
        }
        if (left < best_left) { best_left = left; best.swap(owner); seed_out = seed; }
        if (best_left == 0) break;
    }
    left_out = best_left;
    return best;
}

inline KernelTables build_kernel_tables(const svjg_graph &g) {
    KernelTables kt;
    uint64_t nsz = 16;
    while (nsz < 5 * g.n_nodes / 2 + 2) nsz *= 2;            // load <= 0.4
    kt.names.assign(nsz * NAME_ENT_WORDS, 0);
    for (uint64_t j = 0; j < nsz; ++j) kt.names[j * NAME_ENT_WORDS + 6] = NAME_EMPTY;
    kt.name_mask = (uint32_t)nsz - 1;
    {
        std::vector<uint32_t> ent;                            // 10 words per candidate: d[0..7], meta, len_bp
        std::vector<uint32_t> pre;
        for (uint64_t i = 0; i < g.n_nodes; ++i) {
            const svjg_node &nd = g.nodes[i];
            uint32_t c = (uint32_t)(nd.key >> 48), pos = (uint32_t)(nd.key >> 16), kind = (uint32_t)(nd.key >> 15) & 1u, cnt = (uint32_t)nd.key & 0x7FFFu;
            std::string nm(g.chrom_names + g.chrom_off[c], g.chrom_off[c + 1] - g.chrom_off[c]);
            nm += ":" + std::to_string(pos) + (kind ? "." + std::to_string(cnt) : "-" + std::to_string(nd.aux));
            if (nm.size() > 32 || i > NAME_MAX_ID) { ++kt.names_skipped; continue; }    // such a name can only be handled by the exact path
            uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t b = 0; b < nm.size(); ++b) d[b >> 2] |= (uint32_t)(uint8_t)nm[b] << (8 * (b & 3));
            uint32_t flags = ((nd.row & 0x80000000u) ? 1u : 0u) | ((kind && nd.aux == SVJG_LEN_UNKNOWN) ? 2u : 0u);
            uint32_t len_bp = kind ? nd.aux : nd.aux - pos + 1;
            for (int w = 0; w < 8; ++w) ent.push_back(d[w]);
            ent.push_back(((uint32_t)i << 7) | (flags << 5) | ((uint32_t)nm.size() - 1u));
            ent.push_back(len_bp);
            pre.push_back(name_prehash_host(d, (uint32_t)nm.size()));
        }
        std::vector<int64_t> owner = cuckoo_place(pre, kt.name_mask, kt.name_seed, kt.names_left_out);
        for (uint64_t j = 0; j < nsz; ++j) {
            if (owner[j] < 0) continue;
            const uint32_t *src = &ent[(size_t)owner[j] * 10];
            uint32_t *e = &kt.names[j * NAME_ENT_WORDS];
            for (int w = 0; w < 6; ++w) e[w] = src[w];
            e[6] = src[8]; e[7] = src[9]; e[8] = src[6]; e[9] = src[7];
        }
    }
    uint64_t lsz = 16;
    while (lsz < 5 * g.n_edges / 2 + 2) lsz *= 2;
    kt.links.assign(lsz * LINK_ENT_WORDS, 0xFFFFFFFFu);
    kt.link_mask = (uint32_t)lsz - 1;
    {
        std::vector<uint32_t> ent, pre;                       // 4 words per candidate
        for (uint64_t n = 0; n < g.n_nodes; ++n) {
            uint32_t a = g.nodes[n].row & 0x7FFFFFFFu, b = g.nodes[n + 1].row & 0x7FFFFFFFu;
            for (uint32_t i = a; i < b; ++i) {
                const svjg_edge &ed = g.edges[i];
                uint64_t key = ((uint64_t)n << 33) | ((uint64_t)(ed.meta & 1u) << 32) | ((uint64_t)ed.right << 1) | ((ed.meta >> 1) & 1u);
                const uint32_t nh = ed.meta >> 2;
                ent.push_back((uint32_t)key); ent.push_back((uint32_t)(key >> 32));
                if (nh == 1) { ent.push_back(ed.h0); ent.push_back(LINK_NO_HIT); }
                else if (nh == 2) { ent.push_back(ed.h0); ent.push_back(ed.h1); }
                else { ent.push_back(LINK_MANY | ed.h0); ent.push_back(nh); }
                pre.push_back(link_prehash_host(key));
            }
        }
        std::vector<int64_t> owner = cuckoo_place(pre, kt.link_mask, kt.link_seed, kt.links_left_out);
        for (uint64_t j = 0; j < lsz; ++j) {
            if (owner[j] < 0) continue;
            for (int w = 0; w < 4; ++w) kt.links[j * LINK_ENT_WORDS + w] = ent[(size_t)owner[j] * 4 + w];
        }
    }
    return kt;
}

}  // namespace svjg
