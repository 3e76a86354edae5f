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
// NAME TABLE: canonical node name (<= 32 bytes, zero padded to eight words) -> node.  The kernel hashes the raw
// bytes of a path segment and compares them with the stored spelling: no number parsing on the device, and
// only names spelled exactly like the graph's can match (anything else goes to the exact path).
//   entry = 16 words: [0..7] name, [8] byte length | flags << 8 (bit 0 hazard-prone, bit 1 length unknown),
//                     [9] node id, [10] node length in bp, [11..15] unused.   byte length 0 = empty slot.
//   (A 16-byte fingerprint entry was measured too: 2 % faster, not worth giving up the exact comparison.)
// LINK TABLE: (left id, left strand, right id, right strand) -> hits, same content as the CSR rows.
//   entry = 4 words (one 16-byte load): [0] key low, [1] key high, [2] a, [3] b with
//       1 hit : a = hit, b = LINK_NO_HIT        2 hits: a, b = the hits
//       more  : a = LINK_MANY | index into hits[], b = number of hits
//   key = left << 33 | left strand << 32 | right << 1 | right strand ; all ones = empty slot.
constexpr uint32_t NAME_ENT_WORDS = 16, LINK_ENT_WORDS = 4;
constexpr uint32_t LINK_NO_HIT = 0xFFFFFFFFu, LINK_MANY = 0x80000000u;

inline uint32_t name_hash_host(const uint32_t *d, uint32_t len) {
    static const uint32_t C[8] = {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u};
    uint32_t h = len * 0x7FEB352Du;
    for (int i = 0; i < 8; ++i) h += d[i] * C[i];
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}

inline uint32_t link_hash_host(uint64_t key) {
    uint32_t x = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
    return x;
}

struct KernelTables {
    std::vector<uint32_t> names; uint32_t name_mask = 0;
    std::vector<uint32_t> links; uint32_t link_mask = 0;
};

inline KernelTables build_kernel_tables(const svjg_graph &g) {
    KernelTables kt;
    uint64_t nsz = 16;
    while (nsz < 2 * g.n_nodes + 2) nsz *= 2;
    kt.names.assign(nsz * NAME_ENT_WORDS, 0);
    kt.name_mask = (uint32_t)nsz - 1;
    for (uint64_t i = 0; i < g.n_nodes; ++i) {
        const svjg_node &nd = g.nodes[i];
        uint32_t c = (uint32_t)(nd.key >> 48), pos = (uint32_t)(nd.key >> 16), kind = (uint32_t)(nd.key >> 15) & 1u, cnt = (uint32_t)nd.key & 0x7FFFu;
        std::string nm(g.chrom_names + g.chrom_off[c], g.chrom_off[c + 1] - g.chrom_off[c]);
        nm += ":" + std::to_string(pos) + (kind ? "." + std::to_string(cnt) : "-" + std::to_string(nd.aux));
        if (nm.size() > 32) continue;                       // such a name can only be handled by the exact path
        uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t b = 0; b < nm.size(); ++b) d[b >> 2] |= (uint32_t)(uint8_t)nm[b] << (8 * (b & 3));
        uint32_t flags = ((nd.row & 0x80000000u) ? 1u : 0u) | ((kind && nd.aux == SVJG_LEN_UNKNOWN) ? 2u : 0u);
        uint32_t len_bp = kind ? nd.aux : nd.aux - pos + 1;
        uint64_t j = name_hash_host(d, (uint32_t)nm.size()) & kt.name_mask;
        while (kt.names[j * NAME_ENT_WORDS + 8] & 0xFFu) j = (j + 1) & kt.name_mask;
        uint32_t *e = &kt.names[j * NAME_ENT_WORDS];
        for (int w = 0; w < 8; ++w) e[w] = d[w];
        e[8] = (uint32_t)nm.size() | (flags << 8); e[9] = (uint32_t)i; e[10] = len_bp;
    }
    uint64_t lsz = 16;
    while (lsz < 2 * g.n_edges + 2) lsz *= 2;
    kt.links.assign(lsz * LINK_ENT_WORDS, 0xFFFFFFFFu);
    kt.link_mask = (uint32_t)lsz - 1;
    for (uint64_t n = 0; n < g.n_nodes; ++n) {
        uint32_t a = g.nodes[n].row & 0x7FFFFFFFu, b = g.nodes[n + 1].row & 0x7FFFFFFFu;
        for (uint32_t i = a; i < b; ++i) {
            const svjg_edge &ed = g.edges[i];
            uint64_t key = ((uint64_t)n << 33) | ((uint64_t)(ed.meta & 1u) << 32) | ((uint64_t)ed.right << 1) | ((ed.meta >> 1) & 1u);
            uint64_t j = link_hash_host(key) & kt.link_mask;
            while (kt.links[j * LINK_ENT_WORDS] != 0xFFFFFFFFu || kt.links[j * LINK_ENT_WORDS + 1] != 0xFFFFFFFFu) j = (j + 1) & kt.link_mask;
            uint32_t *e = &kt.links[j * LINK_ENT_WORDS];
            const uint32_t nh = ed.meta >> 2;
            e[0] = (uint32_t)key; e[1] = (uint32_t)(key >> 32);
            if (nh == 1) { e[2] = ed.h0; e[3] = LINK_NO_HIT; }
            else if (nh == 2) { e[2] = ed.h0; e[3] = ed.h1; }
            else { e[2] = LINK_MANY | ed.h0; e[3] = nh; }
        }
    }
    return kt;
}

}  // namespace svjg
