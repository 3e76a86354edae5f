// Host-side table builders shared by the library's graph upload and the host test harness:
// the open-addressing chromosome-name hash table that svjg::chrom_lookup probes (exact path) and the
// name / link hash tables of the main kernel.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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
#include "svjg_line.h"

namespace svjg {

// ---- tables of the main kernel ---------------------------------------------------------------------------------
// The kernel is bound by how many 64-byte lines it pulls from beyond the L2 (random lines of a table of tens of MB come
// at ~66 G lines/s on MI355X, tools/ubench/randread.hip), so the tables are built for ONE line per path node:
// NODE RECORDS: perfect hash (hash and displace, svjg_line.h: name_prehash / name_bucket / name_slot) of the canonical
// node names (<= 48 bytes).  The kernel hashes the raw bytes of a path segment, reads the bucket's 2-byte displacement
// (a small array that stays cached), fetches the one record the name can be in and compares the spelling: no number
// parsing on the device, and only names spelled exactly like the graph's can match (anything else: exact path).
//   record = 16 words (one 64-byte line):
//     [0..5] the name's first three windows (svjg_line.h: name_windows)   [6] node id << 8 | flags << 6 | (byte length - 1)   (flags: bit 0 hazard-prone name,
//     bit 1 length unknown; all ones = empty slot)   [7] node length in bp | REC_ROW_INLINE if the node has no other links
//     than the inline ones
//     names of up to 24 bytes: [8..15] = four inline links;  25..32 bytes: [8..9] = window words 6, 7, [10..15] = three
//     inline links;  33..40 bytes (contig names like chr1_KI270706v1_random): [8..11] = window words 6..9, [12..15] = two inline links
//     (r06: one before);  41..64 bytes: [8..13] = window words 6..11, [14..15] = one inline link.  An inline link = two words: key = right id << 2 | left strand | right strand << 1 (all ones = none),
//     value = the hit (slot << 1 | allele) of a one-hit link, or REC_MANY | index into the inline hit list
//     (ihits[index] = number of hits, then the hits).  Reference-allele links come first.  Nearly every path step is
//     answered from the record that the node lookup fetched anyway and never touches the link table.
// LINK TABLE: (left id, left strand, right id, right strand) -> hits, same content as the CSR rows; two-choice (cuckoo)
// table: every key sits in one of its TWO candidate slots.  The SLOTS of a link are hashed from the 64-bit name pre-hashes of
// its two nodes and the strands (svjg_line.h: link_prehash / link_slots), not from the ids.
//   entry = 4 words (one 16-byte load): [0] key low, [1] key high, [2] a, [3] b with
//       1 hit : a = hit, b = LINK_NO_HIT        2 hits: a, b = the hits
//       more  : a = LINK_MANY | index into hits[], b = number of hits
//   key = left << 33 | left strand << 32 | right << 1 | right strand ; all ones = empty slot.
constexpr uint32_t NAME_ENT_WORDS = 16, LINK_ENT_WORDS = 4;
constexpr uint32_t LINK_NO_HIT = 0xFFFFFFFFu, LINK_MANY = 0x80000000u;
constexpr uint32_t NAME_EMPTY = 0xFFFFFFFFu, NAME_MAX_ID = (1u << 24) - 2u;
static_assert((uint64_t)NAME_MAX_ID + 1 < (1ull << (32 - NAME_ID_SHIFT)), "a record keeps id << NAME_ID_SHIFT in one word, all ones = empty");
constexpr uint32_t REC_ROW_INLINE = 0x80000000u, REC_NO_LINK = 0xFFFFFFFFu, REC_MANY = 0x80000000u;
inline bool name_ent_empty(const uint32_t *e) { return e[6] == NAME_EMPTY; }
inline uint32_t name_ent_len(const uint32_t *e) { return (e[6] & NAME_LEN_MASK) + 1u; }
inline uint32_t name_ent_id(const uint32_t *e) { return e[6] >> NAME_ID_SHIFT; }
inline void name_ent_words(const uint32_t *e, uint32_t d[NAME_WORDS]) {
    for (int w = 0; w < 6; ++w) d[w] = e[w];
    const uint32_t n = name_ent_len(e);
    d[6] = n > 24u ? e[8] : 0u; d[7] = n > 24u ? e[9] : 0u;
    d[8] = n > 32u ? e[10] : 0u; d[9] = n > 32u ? e[11] : 0u; d[10] = n > 40u ? e[12] : 0u; d[11] = n > 40u ? e[13] : 0u;
}
inline uint32_t nm_first_link(uint32_t meta) { const uint32_t n = (meta & NAME_LEN_MASK) + 1u; return n > 40u ? 14u : n > 32u ? 12u : n > 24u ? 10u : 8u; }   // word of the first inline link
inline uint32_t rec_first_link(const uint32_t *e) { return nm_first_link(e[6]); }

struct KernelTables {
    std::vector<uint32_t> names; uint32_t name_slots = 0, name_buckets = 0;
    std::vector<uint16_t> disp;                               // displacement of every bucket of the name hash
    std::vector<uint32_t> ihits;                              // hit lists of inline links with more than one hit: count, hits...
    std::vector<uint32_t> name_pfx;                           // names of 49..64 bytes: their first len - 48 bytes, 4 words per kernel id (empty: no such name)
    std::vector<uint32_t> links; uint32_t link_mask = 0, link_seed = 0;
    uint64_t names_left_out = 0, links_left_out = 0, links_unplaced = 0;   // links_unplaced: their left nodes are flagged for the exact path
    uint32_t names_skipped = 0;                               // node names the table cannot hold (> 32 bytes, id too large)
    std::vector<uint64_t> node_pre; std::vector<uint8_t> node_has;   // per node: name pre-hash, "is in the name table" (table checks)
    std::vector<uint32_t> node_slot;                          // per node: its record (table checks)
    std::vector<uint32_t> kid, node_of_kid;                   // node index <-> the id the kernel sees (walk order: see build_kernel_tables)
};

// Two-choice placement by random-walk eviction.  pre[i] = pre-hash of key i; returns owner[slot] = key index or -1.
// Keys that cannot be placed under any of a few seeds are dropped (counted in left_out).
inline std::vector<int64_t> cuckoo_place(const std::vector<uint64_t> &pre, uint32_t mask, uint32_t &seed_out, uint64_t &left_out,
                                         std::vector<uint32_t> *left_keys = nullptr) {
    std::vector<int64_t> best;
    std::vector<uint32_t> best_keys;
    uint64_t best_left = ~0ull;
    for (uint32_t seed = 0x5bd1e995u, attempt = 0; attempt < 8; ++attempt, seed = seed * 0x9E3779B1u + 0x7F4A7C15u) {
        std::vector<int64_t> owner((size_t)mask + 1, -1);
        uint64_t left = 0, rng = 0x9E3779B97F4A7C15ull ^ seed;
        std::vector<uint32_t> keys_out;
        for (size_t i = 0; i < pre.size(); ++i) {
            int64_t cur = (int64_t)i;
            uint32_t avoid = 0xFFFFFFFFu;
            bool placed = false;
            for (int kick = 0; kick < 512; ++kick) {
                uint32_t s1, s2;
                link_slots(pre[(size_t)cur], seed, mask, s1, s2);
                if (owner[s1] < 0) { owner[s1] = cur; placed = true; break; }
                if (owner[s2] < 0) { owner[s2] = cur; placed = true; break; }
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                uint32_t victim = (rng >> 33) & 1u ? s1 : s2;
                if (victim == avoid) victim = victim == s1 ? s2 : s1;
                std::swap(cur, owner[victim]);
                avoid = victim;
            }
            if (!placed) { ++left; keys_out.push_back((uint32_t)cur); }   // `cur` (whoever was evicted last) stays out
        }
        if (left < best_left) { best_left = left; best.swap(owner); best_keys.swap(keys_out); seed_out = seed; }
        if (best_left == 0) break;
    }
    left_out = best_left;
    if (left_keys) left_keys->swap(best_keys);
    return best;
}

// Hash and displace: keys h[i] (64-bit pre-hashes, distinct) -> a slot of its own for every key.  Buckets are worked
// off largest first; a bucket's displacement is the first value that sends all its keys to free, distinct slots.
// false: some bucket found no displacement below 65536 (the caller retries with more room).
inline bool chd_place(const std::vector<uint64_t> &h, uint32_t n_slots, uint32_t n_buckets, std::vector<uint16_t> &disp, std::vector<uint32_t> &slot_of) {
    std::vector<std::vector<uint32_t>> bk(n_buckets);
    for (uint32_t i = 0; i < h.size(); ++i) bk[name_bucket(h[i], n_buckets)].push_back(i);
    std::vector<uint32_t> order(n_buckets);
    for (uint32_t b = 0; b < n_buckets; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return bk[x].size() > bk[y].size(); });
    std::vector<uint8_t> used(n_slots, 0);
    disp.assign(n_buckets, 0);
    slot_of.assign(h.size(), 0);
    std::vector<uint32_t> tmp;
    for (uint32_t b : order) {
        const std::vector<uint32_t> &keys = bk[b];
        if (keys.empty()) break;
        uint32_t d = 0;
        for (; d < 65536; ++d) {
            tmp.clear();
            bool ok = true;
            for (uint32_t k : keys) {
                const uint32_t s = name_slot(h[k], d, n_slots);
                if (used[s]) { ok = false; break; }
                for (uint32_t t : tmp) if (t == s) { ok = false; break; }
                if (!ok) break;
                tmp.push_back(s);
            }
            if (ok) break;
        }
        if (d == 65536) return false;
        disp[b] = (uint16_t)d;
        for (size_t j = 0; j < keys.size(); ++j) { used[tmp[j]] = 1; slot_of[keys[j]] = tmp[j]; }
    }
    return true;
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
    // The ids the kernel sees follow the order in which a walk along the genome meets the nodes: by chromosome and position, an
    // insertion's node "c:p.n" BEFORE the reference node "c:p-e" that follows it in every path (the sorted node table has it behind:
    // kind is the lower key bit).  The kernel skips its search for repeated names in lines whose ids rise or fall all the way.
    kt.kid.assign((size_t)g.n_nodes, 0); kt.node_of_kid.assign((size_t)g.n_nodes, 0);
    {
        std::vector<uint32_t> ord((size_t)g.n_nodes);
        for (uint32_t i = 0; i < g.n_nodes; ++i) ord[i] = i;
        auto walk_key = [&](uint32_t i) { const uint64_t k = g.nodes[i].key; return (k & ~0xFFFFull) | (((k >> 15) & 1u) ? (k & 0x7FFFu) : 0x8000ull); };
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return walk_key(a) < walk_key(b); });
        for (uint32_t r = 0; r < g.n_nodes; ++r) { kt.kid[ord[r]] = r; kt.node_of_kid[r] = ord[r]; }
    }
    const std::vector<uint32_t> &kid = kt.kid;
    std::vector<uint64_t> node_pre((size_t)g.n_nodes, 0);    // name pre-hash of every node the name table holds
    std::vector<uint8_t> node_has((size_t)g.n_nodes, 0);
    std::vector<uint32_t> node_slot((size_t)g.n_nodes, 0);
    {
        std::vector<uint32_t> ent;                            // NAME_WORDS + 2 words per key: d[0..11], meta, len_bp
        constexpr size_t EW = NAME_WORDS + 2;
        std::vector<uint64_t> hs;
        std::vector<uint32_t> key_node;
        for (uint64_t i = 0; i < g.n_nodes; ++i) {
            const svjg_node &nd = g.nodes[i];
            uint32_t c = (uint32_t)(nd.key >> 48), pos = (uint32_t)(nd.key >> 16), kind = (uint32_t)(nd.key >> 15) & 1u, cnt = (uint32_t)nd.key & 0x7FFFu;
            std::string nm(g.chrom_names + g.chrom_off[c], g.chrom_off[c + 1] - g.chrom_off[c]);
            nm += ":" + std::to_string(pos) + (kind ? "." + std::to_string(cnt) : "-" + std::to_string(nd.aux));
            if (nm.size() > NAME_MAX_BYTES || kid[i] > NAME_MAX_ID) { ++kt.names_skipped; continue; }    // such a name can only be handled by the exact path (the record holds the WALK-ORDER id in 24 bits)
            uint32_t d[NAME_WORDS];
            uint64_t pfx_hash = 0;
            if (nm.size() > 4 * NAME_WORDS) {                 // 49..64 bytes: windows of the last 48, the bytes in front of them in name_pfx
                name_windows(nm.data() + (nm.size() - 4 * NAME_WORDS), 0, 4 * NAME_WORDS, d);
                if (kt.name_pfx.empty()) kt.name_pfx.assign((size_t)g.n_nodes * NAME_PFX_WORDS, 0u);
                uint32_t *pw = &kt.name_pfx[(size_t)kid[i] * NAME_PFX_WORDS];
                name_prefix_words(nm.data(), 0, (uint32_t)nm.size(), pw);
                pfx_hash = name_pfx_hash(pw);
            } else
            name_windows(nm.data(), 0, (uint32_t)nm.size(), d);
            uint32_t flags = ((nd.row & 0x80000000u) ? 1u : 0u) | ((kind && nd.aux == SVJG_LEN_UNKNOWN) ? 2u : 0u);
            uint32_t len_bp = kind ? nd.aux : nd.aux - pos + 1;
            if (len_bp & REC_ROW_INLINE) flags |= 2u;         // (no node is 2 Gbp long; keeps the flag bit free)
            for (uint32_t w = 0; w < NAME_WORDS; ++w) ent.push_back(d[w]);
            ent.push_back((kid[i] << NAME_ID_SHIFT) | (flags << NAME_LEN_BITS) | ((uint32_t)nm.size() - 1u));
            ent.push_back(len_bp & ~REC_ROW_INLINE);
            hs.push_back(name_prehash(d, (uint32_t)nm.size()) + pfx_hash);
            key_node.push_back((uint32_t)i);
        }
        // two names with one 64-bit pre-hash cannot be told apart by any displacement: both stay out (exact path)
        {
            std::vector<uint32_t> idx(hs.size());
            for (uint32_t i = 0; i < idx.size(); ++i) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return hs[x] < hs[y]; });
            std::vector<uint8_t> drop(hs.size(), 0);
            for (size_t i = 1; i < idx.size(); ++i) if (hs[idx[i]] == hs[idx[i - 1]]) { drop[idx[i]] = drop[idx[i - 1]] = 1; }
            size_t w = 0;
            for (size_t i = 0; i < hs.size(); ++i) {
                if (drop[i]) { ++kt.names_left_out; continue; }
                if (w != i) { hs[w] = hs[i]; key_node[w] = key_node[i]; for (size_t q = 0; q < EW; ++q) ent[w * EW + q] = ent[i * EW + q]; }
                ++w;
            }
            hs.resize(w); key_node.resize(w); ent.resize(w * EW);
        }
        lap("name words and pre-hashes");
        const uint64_t n = hs.size();
        std::vector<uint32_t> slot_of;
        uint64_t slots = n + n / 4 + 16;                      // load <= 0.8
        for (int grow = 0;; ++grow) {
            kt.name_slots = (uint32_t)slots;
            // keys per bucket: 8 since r04 (3 before): the displacement array is a third of the size (50 KB at configs[2], 250 KB at
            // configs[3]) and hits the caches more often: -0.9 % kernel time on both (profiles/r04/experiments/displacement_buckets.txt);
            // the largest displacement is ~15 000 of the 65 535 a u16 holds, 12 keys already reach for the limit.  (SVJG_NAME_LAMBDA: measurement only)
            uint32_t lambda = 8;
            { const char *e = getenv("SVJG_NAME_LAMBDA"); if (e && atoi(e) >= 1 && atoi(e) <= 16) lambda = (uint32_t)atoi(e); }
            kt.name_buckets = (uint32_t)(n / lambda + 1);
            if (chd_place(hs, kt.name_slots, kt.name_buckets, kt.disp, slot_of)) break;
            if (grow == 6) { kt.names_left_out += n; hs.clear(); key_node.clear(); slot_of.clear(); break; }   // never seen: everything takes the exact path
            slots += slots / 4;
        }
        lap("hash and displace");
        kt.names.assign((size_t)kt.name_slots * NAME_ENT_WORDS, 0);
        for (uint64_t j = 0; j < kt.name_slots; ++j) {
            uint32_t *e = &kt.names[j * NAME_ENT_WORDS];
            e[6] = NAME_EMPTY; e[8] = e[10] = e[12] = e[14] = REC_NO_LINK;
        }
        for (uint64_t k = 0; k < hs.size(); ++k) {
            const uint32_t *src = &ent[(size_t)k * EW];
            uint32_t *e = &kt.names[(size_t)slot_of[k] * NAME_ENT_WORDS];
            for (int w = 0; w < 6; ++w) e[w] = src[w];
            e[6] = src[NAME_WORDS]; e[7] = src[NAME_WORDS + 1];
            const uint32_t w0 = nm_first_link(src[NAME_WORDS]);
            if (w0 >= 10u) { e[8] = src[6]; e[9] = src[7]; }
            if (w0 >= 12u) { e[10] = src[8]; e[11] = src[9]; }
            if (w0 >= 14u) { e[12] = src[10]; e[13] = src[11]; }
            const uint32_t node = key_node[k];
            node_pre[node] = hs[k]; node_has[node] = 1; node_slot[node] = slot_of[k];
            // inline links: up to two rows of the node, those whose hits are all reference-allele first
            const uint32_t a = g.nodes[node].row & 0x7FFFFFFFu, b = g.nodes[node + 1].row & 0x7FFFFFFFu;
            std::vector<uint32_t> rows;
            for (int pass = 0; pass < 2; ++pass)
                for (uint32_t i = a; i < b; ++i) {
                    const svjg_edge &ed = g.edges[i];
                    const uint32_t nh = ed.meta >> 2;
                    if (!nh || kid[ed.right] > NAME_MAX_ID) continue;
                    bool alt = false;
                    for (uint32_t q = 0; q < nh; ++q) alt |= ((nh <= 2 ? (q ? ed.h1 : ed.h0) : g.hits[ed.h0 + q]) & 1u) != 0;
                    if ((int)alt == pass) rows.push_back(i);
                }
            uint32_t n_live = 0;
            for (uint32_t i = a; i < b; ++i) n_live += (g.edges[i].meta >> 2) != 0;
            const uint32_t cap = (16u - w0) / 2u;
            for (uint32_t w = w0; w < 16; w += 2) e[w] = REC_NO_LINK, e[w + 1] = 0;
            for (size_t q = 0; q < rows.size() && q < cap; ++q) {
                const svjg_edge &ed = g.edges[rows[q]];
                const uint32_t nh = ed.meta >> 2;
                uint32_t *l = e + w0 + 2 * q;
                l[0] = (kid[ed.right] << 2) | (ed.meta & 3u);
                if (nh == 1) l[1] = ed.h0;
                else {
                    l[1] = REC_MANY | (uint32_t)kt.ihits.size();
                    kt.ihits.push_back(nh);
                    for (uint32_t j = 0; j < nh; ++j) kt.ihits.push_back(nh <= 2 ? (j ? ed.h1 : ed.h0) : g.hits[ed.h0 + j]);
                }
            }
            if (n_live <= cap && rows.size() == n_live) e[7] |= REC_ROW_INLINE;
        }
    }
    // A link that cannot be placed (never seen with 64-bit pre-hashes at load <= 0.4) would be a silent miss in the
    // kernel: its left node is flagged hazard-prone instead, which sends the lines that touch it to the exact path.  links_left_out counts links lost for good (never seen: the library then uses the exact path only).
    lap("node records with inline links");
    uint64_t lsz = 16;
    while (lsz < 5 * g.n_edges / 2 + 2) lsz *= 2;
    {
        kt.links.assign(lsz * LINK_ENT_WORDS, 0xFFFFFFFFu);
        kt.link_mask = (uint32_t)lsz - 1;
        std::vector<uint32_t> ent, left_node;                 // 4 words per candidate
        std::vector<uint64_t> pre;
        for (uint64_t n = 0; n < g.n_nodes; ++n) {
            uint32_t a = g.nodes[n].row & 0x7FFFFFFFu, b = g.nodes[n + 1].row & 0x7FFFFFFFu;
            for (uint32_t i = a; i < b; ++i) {
                const svjg_edge &ed = g.edges[i];
                if (!node_has[n] || !node_has[ed.right]) continue;   // a node outside the name table sends its lines to the exact path anyway
                uint64_t key = ((uint64_t)kid[n] << 33) | ((uint64_t)(ed.meta & 1u) << 32) | ((uint64_t)kid[ed.right] << 1) | ((ed.meta >> 1) & 1u);
                const uint32_t nh = ed.meta >> 2;
                ent.push_back((uint32_t)key); ent.push_back((uint32_t)(key >> 32));
                if (nh == 1) { ent.push_back(ed.h0); ent.push_back(LINK_NO_HIT); }
                else if (nh == 2) { ent.push_back(ed.h0); ent.push_back(ed.h1); }
                else { ent.push_back(LINK_MANY | ed.h0); ent.push_back(nh); }
                pre.push_back(link_prehash(node_pre[n], ed.meta & 1u, node_pre[ed.right], (ed.meta >> 1) & 1u));
                left_node.push_back((uint32_t)n);
            }
        }
        lap("link candidates");
        std::vector<uint32_t> unplaced;
        uint64_t n_unplaced = 0;
        std::vector<int64_t> owner = cuckoo_place(pre, kt.link_mask, kt.link_seed, n_unplaced, &unplaced);
        for (uint64_t j = 0; j < lsz; ++j) {
            if (owner[j] < 0) continue;
            for (int w = 0; w < 4; ++w) kt.links[j * LINK_ENT_WORDS + w] = ent[(size_t)owner[j] * 4 + w];
        }
        kt.links_unplaced = n_unplaced;
        for (uint32_t k : unplaced) kt.names[(size_t)node_slot[left_node[k]] * NAME_ENT_WORDS + 6] |= NAME_FLAG_HAZARD;
    }
    lap("link table placement");
    kt.node_pre.swap(node_pre); kt.node_has.swap(node_has); kt.node_slot.swap(node_slot);
    return kt;
}

}  // namespace svjg
