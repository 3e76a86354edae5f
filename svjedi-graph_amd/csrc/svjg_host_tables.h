// Host-side helper shared by the library's graph upload and the host test harness: builds the
// open-addressing chromosome-name hash table that svjg::chrom_lookup probes.
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

namespace svjg {

// Position buckets for the node lookup: for chromosome c, bucket b covers positions [b << shift, (b+1) << shift);
// table[base[c] + b] = index of the first node of c whose pos >= b << shift, with one closing entry per chromosome,
// so a node with position p lives in [table[base+b], table[base+b+1]) for b = p >> shift.
struct BucketTable {
    uint32_t shift = 0;
    std::vector<uint32_t> base;      // n_chrom + 1 (base[c+1] - base[c] - 1 = number of buckets of c)
    std::vector<uint32_t> table;
};

inline BucketTable build_buckets(const svjg_graph &g) {
    BucketTable bt;
    auto pos_of = [&](uint64_t i) { return (uint32_t)((g.nodes[i].key >> 16) & 0xFFFFFFFFull); };
    uint64_t want = 2 * g.n_nodes + 1024;
    for (bt.shift = 0; bt.shift < 32; ++bt.shift) {
        uint64_t tot = 0;
        for (uint32_t c = 0; c < g.n_chrom; ++c) {
            uint32_t lo = g.chrom_node_lo[c], hi = g.chrom_node_lo[c + 1];
            tot += (hi > lo ? ((uint64_t)pos_of(hi - 1) >> bt.shift) + 1 : 0) + 1;
        }
        if (tot <= want) break;
    }
    bt.base.assign(g.n_chrom + 1, 0);
    for (uint32_t c = 0; c < g.n_chrom; ++c) {
        uint32_t lo = g.chrom_node_lo[c], hi = g.chrom_node_lo[c + 1];
        uint32_t nb = hi > lo ? (pos_of(hi - 1) >> bt.shift) + 1 : 0;
        bt.base[c + 1] = bt.base[c] + nb + 1;
    }
    bt.table.assign(bt.base[g.n_chrom] + 1, 0);
    for (uint32_t c = 0; c < g.n_chrom; ++c) {
        uint32_t lo = g.chrom_node_lo[c], hi = g.chrom_node_lo[c + 1];
        uint32_t nb = bt.base[c + 1] - bt.base[c] - 1, i = lo;
        for (uint32_t b = 0; b <= nb; ++b) {
            while (i < hi && (pos_of(i) >> bt.shift) < b) ++i;
            bt.table[bt.base[c] + b] = i;
        }
    }
    return bt;
}

}  // namespace svjg

namespace svjg {

// Word-based chromosome dictionary for the main kernel: names of up to 16 bytes as four little-endian words
// (zero padded) + length, found through an open-addressing table keyed by chrom_word_hash (same function on
// the device, svjg_line.h).  Longer names are left out: alignments that use them take the exact path.
struct ChromWords {
    std::vector<uint32_t> w4;        // n_chrom * 4
    std::vector<uint32_t> table;     // value = chrom index + 1, 0 = empty
};

inline uint32_t chrom_word_hash_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t len) {
    uint32_t h = (c0 * 0x9E3779B1u) ^ (c1 * 0x85EBCA77u) ^ (c2 * 0xC2B2AE3Du) ^ (c3 * 0x27D4EB2Fu) ^ (len * 0x165667B1u);
    return h ^ (h >> 15);
}

inline ChromWords build_chrom_words(const svjg_graph &g) {
    ChromWords cw;
    uint32_t sz = 8;
    while (sz < 2 * g.n_chrom + 2) sz *= 2;
    cw.table.assign(sz, 0);
    cw.w4.assign((size_t)g.n_chrom * 4 + 4, 0);
    for (uint32_t c = 0; c < g.n_chrom; ++c) {
        uint32_t o = g.chrom_off[c], n = g.chrom_off[c + 1] - o;
        if (n == 0 || n > 16) { cw.w4[c * 4] = 0xFFFFFFFFu; continue; }          // never matches a masked name
        uint8_t b[16] = {0};
        for (uint32_t i = 0; i < n; ++i) b[i] = (uint8_t)g.chrom_names[o + i];
        uint32_t w[4];
        for (int i = 0; i < 4; ++i) w[i] = b[4 * i] | (b[4 * i + 1] << 8) | (b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
        for (int i = 0; i < 4; ++i) cw.w4[c * 4 + i] = w[i];
        uint32_t j = chrom_word_hash_host(w[0], w[1], w[2], w[3], n) & (sz - 1);
        while (cw.table[j]) j = (j + 1) & (sz - 1);
        cw.table[j] = c + 1;
    }
    return cw;
}

}  // namespace svjg
