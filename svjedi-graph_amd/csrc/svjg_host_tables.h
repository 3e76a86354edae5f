// Host-side helper shared by the library's graph upload and the host test harness: builds the
// open-addressing chromosome-name hash table that svjg::chrom_lookup probes.
#pragma once
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
