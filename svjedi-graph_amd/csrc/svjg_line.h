// Graph-table lookups shared by the classify kernels, and the EXACT per-line routine (slow_line) that
// evaluates a deferred GAF line with the reference's string semantics, including the
// exception it would raise.  The main kernel (svjg_kernels.h, k_classify_main) handles the regular case —
// canonical node names that exist in the graph, plain decimal columns — and defers anything else; nothing
// is ever guessed.
//
// What this replaces in the reference (/root/reference/filter-alignments.py):
//   read_gaf_line :184-198, extract_nodes :351-373, get_aln_links :200-219, reverse_link :221-225,
//   the d_link_sv probes :141-153, check_bkpt_overlap :258-273, get_node_start/end/len :328-349.
//
// Plain C++ on raw pointers so that tests/hostsim can also compile slow_line with g++ and run it against the
// oracle without a GPU (test harness only; the shipped library has no CPU path).
#pragma once
#include <stdint.h>
#include "../../include/svjg.h"

#ifndef SVJG_HD
#define SVJG_HD __host__ __device__ inline __attribute__((always_inline))
#endif
// Where the exact routine's per-node scratch and its table of path pieces live: LDS in the kernels (they define SVJG_TAB_AS as the LDS
// address space before this header: a plain pointer there is a FLAT pointer, and every read through one takes the long way round), plain
// memory in tests/hostsim.
#ifndef SVJG_TAB_AS
#define SVJG_TAB_AS
#endif
// The exact routine's larger helpers are real calls (r04): inlined into the two exact-path kernels they cost 389 / 390 SGPR spills (saved
// exec masks of deeply nested divergent code); as calls 0 / 8, with a stack of 544 bytes a lane, at the same kernel times
// (profiles/r04/experiments/exact_path.txt).  -DSVJG_SLOW_INLINE builds the inlined form.
#ifndef SVJG_FN
#ifdef SVJG_SLOW_INLINE
#define SVJG_FN SVJG_HD
#else
#define SVJG_FN __host__ __device__ __attribute__((noinline))
#endif
#endif

namespace svjg {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr uint32_t FNV_INIT = 2166136261u;
constexpr uint32_t FNV_PRIME = 16777619u;

struct GraphView {
    const svjg_node *nodes;      // n_nodes + 1 (sentinel)
    uint32_t n_nodes;
    const svjg_edge *edges;
    const uint32_t *hits;
    const uint8_t *chrom_names;  // in LDS inside the classify kernel
    const uint32_t *chrom_off;   // n_chrom + 1
    const uint32_t *chrom_lo;    // n_chrom + 1 : node range of each chromosome
    const uint32_t *chrom_hash;  // open addressing, value = chrom index + 1, 0 = empty
    uint32_t n_chrom;
    uint32_t hash_mask;
    uint32_t d_over;
    uint32_t dover_list;         // SVJG_GRAPH_DOVER_LIST: the comparison with d_over raises TypeError (filter-alignments.py -O)
    const uint32_t *name_ihits;  // hit lists of the records' inline links with more than one hit
    const uint32_t *name_tab;    // canonical node name -> node record (svjg_host_tables.h), 16 words (one 64-byte line) per slot
    const uint16_t *name_disp;   // perfect hash of the node names: displacement of every bucket
    const uint32_t *node_of_kid; // the records hold the main kernel's node ids (walk order, svjg_host_tables.h): id -> index into nodes[] (exact path)
    uint32_t name_slots, name_buckets;
    uint32_t name_complete;      // every node name is in name_tab: a miss there means "no such node" (else: search the sorted table)
    const uint32_t *name_pfx;    // main kernel: the first len - 48 bytes of a name of 49..64 bytes, 4 words per node id (kernel's ids); nullptr: the graph has no such name
    const uint32_t *link_tab;    // main kernel: (left, strand, right, strand) -> hits, 4 words per entry
    uint32_t link_mask, link_seed;
};


// ---- perfect hash of the node names (hash and displace; built by svjg_host_tables.h) -------------------------------
// pre-hash = 64-bit multilinear sum of the name's twelve WINDOW WORDS and its length; bucket from the pre-hash,
// slot from the pre-hash and the bucket's displacement: every name of the graph has a slot of its own, so a lookup
// touches one 2-byte displacement (a small, cache-resident array) and exactly ONE 64-byte record.
//
// Window words (name_windows): a name is hashed and compared through 8-byte windows that END where the name ends, so the
// kernel reads them from the staged text as they are — no masking of the bytes behind a name, which was 30 of the node
// pass's instructions.  For a name of len bytes, o2 = min(max(len - 8, 0), 16) and o1 = o2 / 2:
//     d[0..1] = bytes [0, 8)      d[2..3] = bytes [o1, o1 + 8)      d[4..5] = bytes [o2, o2 + 8)          (they cover a name of 8..24 bytes)
//     d[6..7] = bytes [len - 8, len) if len > 24, else 0                                        (.. of 25..32 bytes)
//     d[8..11] = bytes [len - 24, len - 8) if len > 40                                          (.. of 41..48 bytes)
//     d[8..9]  = bytes [len - 16, len - 8), d[10..11] = 0 if 32 < len <= 40                     (.. of 33..40 bytes: r06 — two words fewer in the
//                record, i.e. room for a second inline link: GRCh38's chr1_KI270706v1_random:1234567-1234999 has 38 bytes)
// Only a name shorter than 8 bytes has bytes behind its end in a window (all three are [0, 8) then): those read as zero.
// (windows, len) determine the name, so comparing them is comparing the spelling.
SVJG_HD uint32_t fmix32(uint32_t z) { z ^= z >> 16; z *= 0x7FEB352Du; z ^= z >> 15; z *= 0x846CA68Bu; z ^= z >> 16; return z; }
SVJG_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
constexpr uint32_t NAME_WORDS = 12;                               // a node name of the main kernel: up to 48 bytes, as window words
constexpr uint32_t NAME_LEN_BITS = 6, NAME_LEN_MASK = 63u;        // record word 6 = id << 8 | flags << 6 | (byte length - 1)
constexpr uint32_t NAME_FLAG_HAZARD = 1u << 6, NAME_FLAG_NOLEN = 1u << 7, NAME_ID_SHIFT = 8;
template <class P>
SVJG_HD void name_windows(P t, uint64_t s, uint32_t len, uint32_t d[NAME_WORDS]) {      // 1 <= len <= 48 bytes at t[s ..]
    const uint32_t o2 = len < 8u ? 0u : (len - 8u < 16u ? len - 8u : 16u), o1 = o2 >> 1;
    auto word = [&](uint32_t at) -> uint32_t {                   // four bytes from name offset `at`, zero behind the name's end
        uint32_t w = 0;
        for (uint32_t b = 0; b < 4; ++b) if (at + b < len) w |= (uint32_t)(uint8_t)t[s + at + b] << (8 * b);
        return w;
    };
    d[0] = word(0); d[1] = word(4); d[2] = word(o1); d[3] = word(o1 + 4); d[4] = word(o2); d[5] = word(o2 + 4);
    d[6] = len > 24u ? word(len - 8u) : 0u; d[7] = len > 24u ? word(len - 4u) : 0u;
    if (len > 40u) for (uint32_t i = 0; i < 4; ++i) d[8 + i] = word(len - 24u + 4u * i);
    else { d[8] = len > 32u ? word(len - 16u) : 0u; d[9] = len > 32u ? word(len - 12u) : 0u; d[10] = 0u; d[11] = 0u; }
}
// r06 — names of 49..64 bytes (contigs named like assemblies name their scaffolds): the twelve window words are those of the name's LAST 48
// bytes, and its first len - 48 bytes (at most 16: four words, zero behind them) wait in a table of their own, four words per node id
// (GraphView::name_pfx; present only if the graph has such a name).  They enter the pre-hash too, so that contigs which differ only at their
// very beginning do not land in one slot.  (windows of the last 48 bytes, the first len - 48 bytes, len) determine the name.
constexpr uint32_t NAME_MAX_BYTES = 64, NAME_PFX_WORDS = 4;
template <class P>
SVJG_HD void name_prefix_words(P t, uint64_t s, uint32_t len, uint32_t p[NAME_PFX_WORDS]) {      // 48 < len <= 64 bytes at t[s ..]
    const uint32_t n = len - 4u * NAME_WORDS;
    for (uint32_t w = 0; w < NAME_PFX_WORDS; ++w) {
        uint32_t v = 0;
        for (uint32_t b = 0; b < 4; ++b) if (4u * w + b < n) v |= (uint32_t)(uint8_t)t[s + 4u * w + b] << (8 * b);
        p[w] = v;
    }
}
SVJG_HD uint64_t name_pfx_hash(const uint32_t p[NAME_PFX_WORDS]) {
    return (uint64_t)p[0] * 0xA24BAED5u + (uint64_t)p[1] * 0x9FB21C65u + (uint64_t)p[2] * 0xE7037ED1u + (uint64_t)p[3] * 0x8EBC6AF1u;
}
SVJG_HD uint64_t name_prehash(const uint32_t d[NAME_WORDS], uint32_t len) {
    const uint32_t C[NAME_WORDS] = {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u,
                                    0x94D049BBu, 0xBF58476Du, 0x2545F491u, 0x9FB21C65u};
    uint64_t h = (uint64_t)len * 0x7FEB352Du;
    for (uint32_t i = 0; i < NAME_WORDS; ++i) h += (uint64_t)d[i] * C[i];
    return h;
}
// The pre-hash is a multiply-and-add with 32-bit factors: its words are only lightly mixed (names that differ in a digit or
// two have nearby high words), so bucket and slot each put it through one multiply / xor-shift round of their own (measured:
// without it hash-and-displace finds no placement for a DEL-only single-chromosome graph; two full finalisers — what this
// replaced — cost about three times the instructions of the whole lookup's compare).
SVJG_HD uint32_t name_bucket(uint64_t h, uint32_t n_buckets) {
    uint32_t z = ((uint32_t)h ^ (uint32_t)(h >> 32)) * 0x9E3779B1u;
    z ^= z >> 15;
    return mulhi32(z * 0x846CA68Bu, n_buckets);
}
SVJG_HD uint32_t name_slot(uint64_t h, uint32_t disp, uint32_t n_slots) {
    const uint32_t x = ((uint32_t)h * 0x85EBCA77u) ^ (uint32_t)(h >> 32);
    uint32_t y = x * ((disp * 0x632BE5ABu + 0x7FEB352Du) | 1u);
    y ^= y >> 15;
    return mulhi32(y * 0x2C1B3C6Du, n_slots);
}
// Link table (two-choice): the slots of a link follow from the 64-bit name pre-hashes of its two nodes and the strands
// (1 = '-'), so a lookup needs no node ids; with 64 bits no three links share their pair of slots.
SVJG_HD uint64_t link_prehash(uint64_t hl, uint32_t sl, uint64_t hr, uint32_t sr) {
    return hl * 0x9E3779B97F4A7C15ull + (hr + sl * 0x68E31DA4B5297A4Dull + sr * 0xD6E8FEB86659FD93ull) * 0xC2B2AE3D27D4EB4Full;
}
SVJG_HD void link_slots(uint64_t v, uint32_t seed, uint32_t mask, uint32_t &s1, uint32_t &s2) {
    s1 = fmix32((uint32_t)v ^ seed) & mask;
    s2 = fmix32((uint32_t)(v >> 32) + seed * 0x85EBCA6Bu) & mask;
    if (s2 == s1) s2 = s1 ^ 1u;
}

// Two blank sets (r06; the r05 tree used the first for both and accepted "1100\x1f" as a column where the reference dies):
// str.rstrip() / str.strip() without an argument take what str.isspace() takes — for ASCII: ' ', 9..13 AND 28..31 (filter-alignments.py:125);
// int() / float() of a str strip what C's isspace() takes — ' ', 9..13 only: 0x1C..0x1F next to a number is a ValueError (:188-194).
SVJG_HD bool py_strip_space(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }
SVJG_HD bool c_space(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// ---- byte searches of the exact routine, eight bytes per step ---------------------------------------------------------------
// The exact routine walks a line byte by byte; where it only LOOKS FOR a byte (the next tab, the next path separator, the next 'i' of
// an "id:f:") the text is taken eight bytes at a time (one unaligned 64-bit read: LDS and global memory of gfx950 serve any byte
// address) and the byte is found with the zero-byte trick, whose lowest flag is exact.  t[q, e) must be readable, nothing beyond.
typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
template <class P>
SVJG_HD uint64_t ld64(P t, uint64_t q) { return *(const u64_unaligned *)(&t[q]); }
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
SVJG_HD uint64_t ld64(const __attribute__((address_space(3))) uint8_t *t, uint64_t q) { return *(const __attribute__((address_space(3))) u64_unaligned *)(t + q); }
#endif
SVJG_HD uint64_t eq_byte_flags(uint64_t w, uint32_t c) {            // 0x80 in the lowest byte of w that equals c (flags above it may be wrong)
    const uint64_t x = w ^ (0x0101010101010101ull * c);
    return (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
}
template <class P>
SVJG_HD uint64_t find_byte(P t, uint64_t q, uint64_t e, uint32_t c) {   // first position in [q, e) that holds c, or e
    for (; q + 8 <= e; q += 8) { const uint64_t m = eq_byte_flags(ld64(t, q), c); if (m) return q + ((uint64_t)__builtin_ctzll(m) >> 3); }
    for (; q < e; ++q) if ((uint32_t)(uint8_t)t[q] == c) return q;
    return e;
}
template <class P>
SVJG_HD uint64_t find_byte2(P t, uint64_t q, uint64_t e, uint32_t c1, uint32_t c2) {   // ... that holds c1 or c2
    for (; q + 8 <= e; q += 8) {
        const uint64_t w = ld64(t, q), m1 = eq_byte_flags(w, c1), m2 = eq_byte_flags(w, c2);
        if (m1 | m2) { const uint64_t a = m1 ? (uint64_t)__builtin_ctzll(m1) : 64u, b = m2 ? (uint64_t)__builtin_ctzll(m2) : 64u; return q + ((a < b ? a : b) >> 3); }
    }
    for (; q < e; ++q) { const uint32_t c = (uint8_t)t[q]; if (c == c1 || c == c2) return q; }
    return e;
}

// ---------------------------------------------------------------------------------------------------
// graph lookups
// ---------------------------------------------------------------------------------------------------

template <class P>
SVJG_HD uint32_t chrom_lookup(const GraphView &g, P t, uint64_t s, uint32_t len, uint32_t h) {
    uint32_t i = h & g.hash_mask;
    for (;;) {
        uint32_t v = g.chrom_hash[i];
        if (v == 0) return NONE32;
        uint32_t c = v - 1, o = g.chrom_off[c];
        if (g.chrom_off[c + 1] - o == len) {
            uint32_t j = 0;
            while (j < len && g.chrom_names[o + j] == t[s + j]) ++j;
            if (j == len) return c;
        }
        i = (i + 1) & g.hash_mask;
    }
}

SVJG_HD uint64_t node_key(uint32_t cidx, uint32_t pos, uint32_t kind, uint32_t cnt) {
    return ((uint64_t)cidx << 48) | ((uint64_t)pos << 16) | ((uint64_t)kind << 15) | cnt;
}

SVJG_HD uint32_t node_search(const GraphView &g, uint32_t cidx, uint64_t key) {
    uint32_t lo = g.chrom_lo[cidx], hi = g.chrom_lo[cidx + 1];
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        uint64_t k = g.nodes[mid].key;
        if (k < key) lo = mid + 1; else hi = mid;
    }
    return (lo < g.chrom_lo[cidx + 1] && g.nodes[lo].key == key) ? lo : NONE32;
}

// entry of the directed link (l, sl) -> (r, sr), or NONE32
SVJG_HD uint32_t edge_find(const GraphView &g, uint32_t l, uint32_t sl, uint32_t r, uint32_t sr) {
    uint32_t a = g.nodes[l].row & 0x7FFFFFFFu, b = g.nodes[l + 1].row & 0x7FFFFFFFu;
    uint32_t want = sl | (sr << 1);
    for (uint32_t i = a; i < b; ++i)
        if (g.edges[i].right == r && (g.edges[i].meta & 3u) == want) return i;
    return NONE32;
}

SVJG_HD uint32_t edge_hit(const GraphView &g, const svjg_edge &e, uint32_t j) {
    uint32_t nh = e.meta >> 2;
    if (nh <= 2) return j == 0 ? e.h0 : e.h1;
    return g.hits[e.h0 + j];
}

// ---------------------------------------------------------------------------------------------------
// exact path
// ---------------------------------------------------------------------------------------------------

// Python int() of t[a, b): blanks (C isspace), sign, digits with single inner underscores (ASCII subset).
//   PY_INT_BAD  : int() raises ValueError (if the text holds bytes >= 0x80 the host asks Python: Unicode digits and blanks)
//   PY_INT_OK   : the value, exactly
//   PY_INT_BIG  : a valid spelling of more than max_digits digits.  Python's integers have no width — and since CPython 3.10.7 int() refuses
//                 more than sys.get_int_max_str_digits() = 4300 digits unless the interpreter is told otherwise —; this routine's are
//                 64 bits wide.  Such a value is never guessed at: the line goes to the host (SVJG_EXC_ASK_HOST), where Python's own int()
//                 decides and the three columns that matter are rewritten within range with the same meaning (svjg/filter.py: host_line).
// max_digits: 18 for a column (|v| < 10^18: Tlen - Te - 1 and the differences with the path sums below stay inside 64 bits), 12 for a
// coordinate inside a node name that is no node of the graph (:343-349: its "length" is end - start + 1, and up to 2^16 of them are summed).
constexpr int PY_INT_BAD = 0, PY_INT_OK = 1, PY_INT_BIG = 2;
constexpr uint32_t COL_DIGITS = 18, NAME_DIGITS = 12, MAX_PATH_NODES = 65536;
template <class P>
SVJG_FN int py_int(P t, uint64_t a, uint64_t b, int64_t &out, uint32_t max_digits = COL_DIGITS) {
    while (a < b && c_space(t[a])) ++a;
    while (b > a && c_space(t[b - 1])) --b;
    bool neg = false;
    if (a < b && (t[a] == '+' || t[a] == '-')) { neg = t[a] == '-'; ++a; }
    if (a >= b || (uint32_t)t[a] - '0' > 9) return PY_INT_BAD;
    int64_t v = 0; bool us = false; uint32_t nd = 0;
    for (; a < b; ++a) {
        uint32_t c = t[a];
        if (c == '_') { if (us) return PY_INT_BAD; us = true; continue; }
        if (c - '0' > 9) return PY_INT_BAD;
        us = false;
        if (nd < max_digits) v = v * 10 + (int64_t)(c - '0');
        if (nd <= max_digits) ++nd;
    }
    if (us) return PY_INT_BAD;
    out = neg ? -v : v;
    return nd > max_digits ? PY_INT_BIG : PY_INT_OK;
}

template <class P>
SVJG_HD uint64_t digits_us(P t, uint64_t i, uint64_t n, bool &ok) {
    uint64_t st = i; bool us = true;
    while (i < n) {
        uint32_t c = t[i];
        if (c - '0' <= 9) { us = false; ++i; }
        else if (c == '_' && !us) { us = true; ++i; }
        else break;
    }
    if (i > st && us) ok = false;
    return i;
}

template <class P>
SVJG_HD bool word_is(P t, uint64_t a, uint64_t b, const char *w, uint32_t wl) {
    if (b - a != wl) return false;
    for (uint32_t i = 0; i < wl; ++i) { uint32_t c = t[a + i]; if (c >= 'A' && c <= 'Z') c += 32; if (c != (uint32_t)w[i]) return false; }
    return true;
}

// would Python's float() accept t[a,b) ?
template <class P>
SVJG_FN bool py_float_ok(P t, uint64_t a, uint64_t b) {
    while (a < b && c_space(t[a])) ++a;
    while (b > a && c_space(t[b - 1])) --b;
    if (a < b && (t[a] == '+' || t[a] == '-')) ++a;
    if (word_is(t, a, b, "inf", 3) || word_is(t, a, b, "infinity", 8) || word_is(t, a, b, "nan", 3)) return true;
    bool ok = true;
    uint64_t i = digits_us(t, a, b, ok); uint64_t nd = i - a;
    if (i < b && t[i] == '.') { uint64_t j = digits_us(t, i + 1, b, ok); nd += j - (i + 1); i = j; }
    if (!ok || nd == 0) return false;
    if (i < b && (t[i] == 'e' || t[i] == 'E')) {
        ++i;
        if (i < b && (t[i] == '+' || t[i] == '-')) ++i;
        uint64_t j = digits_us(t, i, b, ok);
        if (j == i || !ok) return false;
        i = j;
    }
    return i == b;
}

template <class P>
SVJG_HD bool bytes_eq(P t, uint64_t a, uint64_t b, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) if (t[a + i] != t[b + i]) return false;
    return true;
}

// t[a, a + m) == t[b, b + m), eight bytes per step, for callers that have ALREADY found t[a + m, a + m + 8) == t[b + m, b + m + 8): the last
// step may reach into those eight bytes (it reads nothing behind them)
template <class P>
SVJG_HD bool bytes_eq_before_tail(P t, uint64_t a, uint64_t b, uint64_t m) {
    for (uint64_t i = 0; i < m; i += 8) if (ld64(t, a + i) != ld64(t, b + i)) return false;
    return true;
}

// some byte >= 0x80 in t[a, b): what failed the ASCII rules may pass Python's (Unicode digits, Unicode blanks): the host decides
template <class P>
SVJG_HD bool has_high(P t, uint64_t a, uint64_t b) {
    for (uint64_t q = a; q < b; ++q) if ((uint32_t)t[q] >= 0x80u) return true;
    return false;
}

struct NameRef { uint64_t s, e; };     // node name = t[s, e)

// Node names of the path t[ps, pe), one after the other.  oriented: non-empty pieces between '<' / '>' ; otherwise the
// comma-separated pieces minus their last character (filter-alignments.py:366-371).  `pos` starts at ps and is
// advanced past the returned piece; false when the path is exhausted.
template <class P>
SVJG_HD bool next_node(P t, uint64_t pe, bool oriented, uint64_t &pos, NameRef &out) {
    uint64_t st = pos;
    for (;;) {
        const uint64_t q = oriented ? find_byte2(t, st, pe, '<', '>') : find_byte(t, st, pe, ',');   // the piece's end: a separator, or the path's end
        if (q > st) { out.s = st; out.e = oriented ? q : q - 1; pos = q + 1; return true; }
        if (q >= pe) break;
        st = q + 1;
    }
    pos = pe + 1;
    return false;
}

// name_windows for a name INSIDE a line whose twelve columns were found (the exact routine): at least eight more bytes of the line follow
// the name's first byte — six more columns —, so the windows are read eight bytes at a time and only a name shorter than eight bytes is
// masked.  Same words as name_windows (checked by the host build on every fixture: both resolve every node of the graph).
template <class P>
SVJG_HD void name_windows_inline(P t, uint64_t s, uint32_t len, uint32_t d[NAME_WORDS]) {
    const uint32_t o2 = len < 8u ? 0u : (len - 8u < 16u ? len - 8u : 16u), o1 = o2 >> 1;
    uint64_t w0 = ld64(t, s), w1 = ld64(t, s + o1), w2 = ld64(t, s + o2);
    if (len < 8u) { const uint64_t m = ~(~0ull << (8u * len)); w0 &= m; w1 &= m; w2 &= m; }
    d[0] = (uint32_t)w0; d[1] = (uint32_t)(w0 >> 32); d[2] = (uint32_t)w1; d[3] = (uint32_t)(w1 >> 32); d[4] = (uint32_t)w2; d[5] = (uint32_t)(w2 >> 32);
    const uint64_t w3 = len > 24u ? ld64(t, s + len - 8u) : 0ull;
    d[6] = (uint32_t)w3; d[7] = (uint32_t)(w3 >> 32);
    const uint64_t w4 = len > 40u ? ld64(t, s + len - 24u) : len > 32u ? ld64(t, s + len - 16u) : 0ull, w5 = len > 40u ? ld64(t, s + len - 16u) : 0ull;
    d[8] = (uint32_t)w4; d[9] = (uint32_t)(w4 >> 32); d[10] = (uint32_t)w5; d[11] = (uint32_t)(w5 >> 32);
}

// Node-name table of the main kernel (svjg_host_tables.h), probed with the raw bytes of a name of 1..48 bytes:
// node id, or NONE32 when the name's slot does not hold this spelling.
template <class P>
SVJG_HD uint32_t name_tab_find(const GraphView &g, P t, NameRef nm, int64_t *len_bp = nullptr) {   // len_bp: the node's length from its record, or -1 (none there)
    const uint32_t len = (uint32_t)(nm.e - nm.s);
    uint32_t d[NAME_WORDS];
    name_windows_inline(t, nm.s, len, d);
    const uint64_t h = name_prehash(d, len);
    const uint32_t slot = name_slot(h, g.name_disp[name_bucket(h, g.name_buckets)], g.name_slots);
    // the record: 64 bytes, 16-byte aligned, read as four 16-byte words (one load each, like the main kernel's) instead of word by word
    typedef uint32_t w4 __attribute__((vector_size(16)));
    const w4 *e = (const w4 *)(g.name_tab + (uint64_t)slot * 16);
    const w4 a = e[0], b = e[1];
    const uint32_t meta = b[2];
    if (meta == 0xFFFFFFFFu || (meta & NAME_LEN_MASK) != len - 1u) return NONE32;
    if (a[0] != d[0] || a[1] != d[1] || a[2] != d[2] || a[3] != d[3] || b[0] != d[4] || b[1] != d[5]) return NONE32;
    if (len > 24u) {
        const w4 c = e[2];
        if (c[0] != d[6] || c[1] != d[7]) return NONE32;
        if (len > 32u) {
            if (c[2] != d[8] || c[3] != d[9]) return NONE32;
            if (len > 40u) { const w4 f = e[3]; if (f[0] != d[10] || f[1] != d[11]) return NONE32; }   // (33..40 bytes: those two words hold a link)
        }
    }
    if (len_bp) *len_bp = (meta & NAME_FLAG_NOLEN) ? -1 : (int64_t)(b[3] & 0x7FFFFFFFu);      // (record word 7: length in bp | "no other links" << 31)
    return g.node_of_kid[meta >> NAME_ID_SHIFT];
}

// exact name -> node id (only canonical spellings can be in the table)
template <class P>
SVJG_FN uint32_t resolve_name(const GraphView &g, P t, NameRef nm, bool *is_alt_form, int64_t *tab_len = nullptr) {
    if (tab_len) *tab_len = -1;
    uint64_t colon = nm.e;
    for (uint64_t q = nm.e; q > nm.s; --q) if (t[q - 1] == ':') { colon = q - 1; break; }
    if (is_alt_form) {
        *is_alt_form = false;
        for (uint64_t q = (colon == nm.e ? nm.s : colon + 1); q < nm.e; ++q) if (t[q] == '.') *is_alt_form = true;
    }
    if (colon == nm.e) return NONE32;
    if (g.name_tab && nm.e - nm.s <= 4 * NAME_WORDS) {                             // the canonical spelling is the only one that resolves
        uint32_t id = name_tab_find(g, t, nm, tab_len);
        if (id != NONE32 || g.name_complete) return id;
    }
    uint32_t h = FNV_INIT;
    for (uint64_t q = nm.s; q < colon; ++q) h = (h ^ (uint32_t)t[q]) * FNV_PRIME;
    uint32_t cidx = chrom_lookup(g, t, nm.s, (uint32_t)(colon - nm.s), h);
    if (cidx == NONE32) return NONE32;
    uint64_t v[2] = {0, 0}; uint32_t nd[2] = {0, 0}, part = 0, sep = 0;
    for (uint64_t q = colon + 1; q < nm.e; ++q) {
        uint32_t c = t[q], d = c - '0';
        if (d <= 9) {
            if (nd[part] == 1 && v[part] == 0) return NONE32;
            v[part] = v[part] * 10 + d;
            if (++nd[part] > 10) return NONE32;
        } else if ((c == '-' || c == '.') && part == 0 && nd[0]) { sep = c; part = 1; }
        else return NONE32;
    }
    if (!sep || !nd[1] || v[0] > 0xFFFFFFFFull || v[1] > 0xFFFFFFFFull) return NONE32;
    uint32_t kind = sep == '.';
    if (kind && v[1] >= 32768) return NONE32;
    uint32_t id = node_search(g, cidx, node_key(cidx, (uint32_t)v[0], kind, kind ? (uint32_t)v[1] : 0));
    if (id == NONE32) return NONE32;
    if (!kind && g.nodes[id].aux != (uint32_t)v[1]) return NONE32;
    return id;
}

// get_node_len (filter-alignments.py:343-349): 0 = ok, else the exception class.  (id, alt, tab_len): what resolve_name said about the
// name.  tab_len >= 0: the name is a node's one canonical spelling and its record holds the length — for a reference node exactly what
// int(end) - int(start) + 1 gives on that spelling, without reading the name again.
template <class P>
SVJG_HD int node_len_resolved(const GraphView &g, P t, NameRef nm, uint32_t id, bool alt, int64_t &len, int64_t tab_len = -1) {
    if (!alt && id != NONE32 && tab_len >= 0) { len = tab_len; return 0; }
    if (alt) {
        if (id == NONE32 || g.nodes[id].aux == SVJG_LEN_UNKNOWN || !((g.nodes[id].key >> 15) & 1)) return SVJG_EXC_KEY_ERROR;
        len = g.nodes[id].aux;
        return 0;
    }
    uint64_t c0 = nm.s;
    for (uint64_t q = nm.e; q > nm.s; --q) if (t[q - 1] == ':') { c0 = q; break; }
    uint64_t d1 = nm.e;
    for (uint64_t q = c0; q < nm.e; ++q) if (t[q] == '-') { d1 = q; break; }
    if (d1 == nm.e) return SVJG_EXC_INDEX_ERROR;
    uint64_t d2 = nm.e;
    for (uint64_t q = d1 + 1; q < nm.e; ++q) if (t[q] == '-') { d2 = q; break; }
    int64_t a, b;
    // (a coordinate beyond NAME_DIGITS digits: the host refuses the line — svjg/filter.py: host_line, DESIGN §8 —; with bytes >= 0x80 Python's
    //  int() might take Unicode digits where this one fails: the host's too)
    { const int r = py_int(t, d1 + 1, d2, b, NAME_DIGITS); if (r != PY_INT_OK) return r == PY_INT_BIG || has_high(t, d1 + 1, d2) ? SVJG_EXC_ASK_HOST : SVJG_EXC_VALUE_ERROR; }
    { const int r = py_int(t, c0, d1, a, NAME_DIGITS); if (r != PY_INT_OK) return r == PY_INT_BIG || has_high(t, c0, d1) ? SVJG_EXC_ASK_HOST : SVJG_EXC_VALUE_ERROR; }
    len = b - a + 1;
    return 0;
}
template <class P>
SVJG_HD int generic_node_len(const GraphView &g, P t, NameRef nm, int64_t &len) {
    bool alt;
    int64_t tl;
    const uint32_t id = resolve_name(g, t, nm, &alt, &tl);
    return node_len_resolved(g, t, nm, id, alt, len, tl);
}

// one candidate position of the name: does the name lie at q?
template <class P>
SVJG_HD bool name_at(P t, uint64_t q, NameRef nm) {
    const uint64_t n = nm.e - nm.s;
    if (n >= 8) return ld64(t, q + n - 8) == ld64(t, nm.e - 8) && bytes_eq_before_tail(t, q, nm.s, n - 8);
    return bytes_eq(t, q, nm.s, n);
}

// char before the first occurrence of the name as a substring of the path (filter-alignments.py:206)
template <class P>
SVJG_FN int strand_of(P t, uint64_t ps, uint64_t pe, NameRef nm, uint32_t &strand) {
    uint64_t n = nm.e - nm.s;
    if (n == 0) return SVJG_EXC_VALUE_ERROR;                      // str.split("")
    {
        // a name with a ':' lies only where its last ':' meets a ':' of the path: the path's colons, in order (found eight bytes per step),
        // give the candidate positions in order — one per node in front of the name instead of one per byte
        uint64_t cn = n;
        for (uint64_t b = n; b; --b) if (t[nm.s + b - 1] == ':') { cn = b - 1; break; }
        if (cn < n) {
            for (uint64_t c = find_byte(t, ps, pe, ':'); c < pe; c = find_byte(t, c + 1, pe, ':')) {
                if (c < ps + cn) continue;
                const uint64_t q = c - cn;
                if (q + n > pe) break;
                if (name_at(t, q, nm)) {
                    if (q == ps) return SVJG_EXC_INDEX_ERROR;     // ""[-1]
                    strand = t[q - 1] == '>' ? 0u : 1u;
                    return 0;
                }
            }
            return SVJG_EXC_INDEX_ERROR;                          // unreachable: the name is part of the path
        }
    }
    if (n >= 8) {
        // names of one path begin alike (the chromosome) and differ at their ends (the coordinates): a candidate position is first held
        // against the name's LAST eight bytes, one 8-byte read, and only then against the rest
        const uint64_t tail = ld64(t, nm.e - 8);
        for (uint64_t q = ps; q + n <= pe; ++q)
            if (ld64(t, q + n - 8) == tail && bytes_eq_before_tail(t, q, nm.s, n - 8)) {
                if (q == ps) return SVJG_EXC_INDEX_ERROR;         // ""[-1]
                strand = t[q - 1] == '>' ? 0u : 1u;
                return 0;
            }
        return SVJG_EXC_INDEX_ERROR;                              // unreachable: the name is part of the path
    }
    for (uint64_t q = ps; q + n <= pe; ++q)
        if (t[q] == t[nm.s] && bytes_eq(t, q, nm.s, n)) {
            if (q == ps) return SVJG_EXC_INDEX_ERROR;             // ""[-1]
            strand = t[q - 1] == '>' ? 0u : 1u;
            return 0;
        }
    return SVJG_EXC_INDEX_ERROR;                                  // unreachable: the name is part of the path
}

// The same through a table of the path's pieces (the maximal runs of bytes between its separators: '<' '>' of an oriented path, ',' of
// an unoriented one; pos[i] = start | length << 16, offsets below 65536): a node name holds none of its path's separators, so an
// occurrence lies inside ONE piece, and the first occurrence is the first piece — in order — that holds the name, at its first offset.
// Compared from the name's end (names of one path differ in their coordinates).  O(pieces) instead of O(bytes of the path) per name:
// what makes a path of hundreds of nodes affordable for the one-wave-per-line kernel.
// colon[i]: where piece i has its ':' — offset of its only one from the piece's start | 1 << 14; 0: it has none; 2 << 14: several (or a
// piece of 16 KB and more): every offset of such a piece is tried (piece_colons).  key[i]: for a piece with one ':', the eight bytes
// around it, t[colon - 3, colon + 5) (piece_key).  A name with a ':' can only lie where its last ':' meets one of the piece's: no ':'
// in the piece, no occurrence; one, ONE offset — and the name lies there only if the bytes around the two colons agree, which is asked
// first: names of one path differ in their coordinates' leading digits or in their contig, so a pass over the table (three small reads a
// piece) replaces (piece length - name length + 1) string compares per piece.
template <class P>
SVJG_HD uint16_t piece_colons(P t, uint64_t a, uint64_t L) {
    uint32_t cnt = 0, last = 0;
    for (uint64_t q = a; q < a + L; ++q) if (t[q] == ':') { ++cnt; last = (uint32_t)(q - a); }
    return (uint16_t)(cnt == 0 ? 0u : (cnt == 1 && L < 16384u) ? (last | (1u << 14)) : (2u << 14));
}
template <class P>
SVJG_HD uint64_t piece_key(P t, uint64_t a, uint32_t cw) { return (cw >> 14) == 1u ? ld64(t, a + (cw & 0x3FFFu) - 3) : 0ull; }   // (a piece starts behind five columns: no underflow)

// The strand of node j of the path (its name nm begins piece j): the char in front of the name's first occurrence (see above), which is
// in one of the pieces in front of piece j, or else piece j's own start.
template <class P>
SVJG_FN int strand_of_pieces(P t, uint64_t ps, const SVJG_TAB_AS uint32_t *pos, const SVJG_TAB_AS uint16_t *colon, const SVJG_TAB_AS uint64_t *key,
                             uint32_t j, NameRef nm, uint32_t &strand) {
    const uint64_t n = nm.e - nm.s;
    if (n == 0) return SVJG_EXC_VALUE_ERROR;                      // str.split("")
    uint64_t cn = n;                                              // offset of the name's last ':' (n: it has none, or there is no table)
    if (colon && key) for (uint64_t b = n; b; --b) if (t[nm.s + b - 1] == ':') { cn = b - 1; break; }
    uint64_t found = pos[j] & 0xFFFFu;
    bool hit = false;
    if (cn < n) {
        uint64_t mask = 0;                                        // the bytes of the window t[colon - 3, colon + 5) that lie inside the name
        for (uint32_t r = 0; r < 8; ++r) if (cn + r >= 3 && cn + r < n + 3) mask |= 0xFFull << (8 * r);
        const uint64_t kn = ld64(t, nm.s + cn - 3) & mask;
        for (uint32_t i = 0; i < j && !hit; ++i) {
            const uint32_t cw = colon[i], pl = pos[i];            // (three reads that do not wait for each other)
            const uint64_t ky = key[i];
            const uint64_t a = pl & 0xFFFFu, L = pl >> 16, cp = cw & 0x3FFFu;
            const bool one = (cw >> 14) == 1u && cp >= cn && L - cp >= n - cn && (ky & mask) == kn;
            const bool many = (cw >> 14) == 2u && L >= n;
            if (!(one || many)) continue;
            if (one) { if (name_at(t, a + cp - cn, nm)) { found = a + cp - cn; hit = true; } }
            else for (uint64_t q = a; q + n <= a + L && !hit; ++q) if (name_at(t, q, nm)) { found = q; hit = true; }
        }
    } else
        for (uint32_t i = 0; i < j && !hit; ++i) {
            const uint64_t a = pos[i] & 0xFFFFu, L = pos[i] >> 16;
            if (L < n) continue;
            for (uint64_t q = a; q + n <= a + L && !hit; ++q) if (name_at(t, q, nm)) { found = q; hit = true; }
        }
    if (found == ps) return SVJG_EXC_INDEX_ERROR;                 // ""[-1]
    strand = t[found - 1] == '>' ? 0u : 1u;
    return 0;
}

// The per-line part of the exact routine (filter-alignments.py:184-198 read_gaf_line, :351-373 extract_nodes): columns,
// the nine int() columns, the id:f: tag, the path column and its node count.  0 or the exception class.
struct SlowLine { uint64_t ps, pe; bool oriented; uint32_t k; int64_t Tlen, Ts, Te; };

template <class P>
SVJG_FN int slow_prologue(P t, uint64_t s, uint64_t e, SlowLine &o) {
    o.k = 0;
    while (e > s && py_strip_space(t[e - 1])) --e;
    // (str.rstrip() also takes Unicode blanks: a line that ends in a byte >= 0x80 is the host's if anything below fails on it)
    uint64_t fs[12], fe[12]; uint32_t nf = 0;
    { uint64_t st = s;
      while (nf < 12) {
          const uint64_t q = find_byte(t, st, e, '\t');             // (the last field ends with the line)
          fs[nf] = st; fe[nf] = q; ++nf;
          if (q >= e) break;
          st = q + 1;
      } }
    if (nf < 12) return SVJG_EXC_VALUE_ERROR;
    int64_t v[12];
    const int cols[9] = {1, 2, 3, 6, 7, 8, 9, 10, 11};
    for (int j = 0; j < 9; ++j)
    {   const int r = py_int(t, fs[cols[j]], fe[cols[j]], v[cols[j]]);
        if (r != PY_INT_OK) return r == PY_INT_BIG || has_high(t, fs[cols[j]], fe[cols[j]]) ? SVJG_EXC_ASK_HOST : SVJG_EXC_VALUE_ERROR; }
    { uint64_t last = e;                                           // "id:f:" in line  (:193-196)
      for (uint64_t q = find_byte(t, s, e, 'i'); q + 5 <= e; q = find_byte(t, q + 1, e, 'i'))
          if (t[q + 1] == 'd' && t[q + 2] == ':' && t[q + 3] == 'f' && t[q + 4] == ':') last = q;
      if (last != e) {
          uint64_t a = last + 5, b = a;
          while (b < e && t[b] != '\t') ++b;
          if (!py_float_ok(t, a, b)) return has_high(t, a, b) ? SVJG_EXC_ASK_HOST : SVJG_EXC_VALUE_ERROR;
      } else if (v[10] == 0) return SVJG_EXC_ZERO_DIVISION; }
    o.ps = fs[5]; o.pe = fe[5];
    if (o.pe == o.ps) return SVJG_EXC_INDEX_ERROR;                 // p[0]
    o.oriented = t[o.ps] == '<' || t[o.ps] == '>';
    { NameRef nm{0, 0}; uint64_t pos = o.ps; while (next_node(t, o.pe, o.oriented, pos, nm)) ++o.k; }
    if (o.k > MAX_PATH_NODES) return SVJG_EXC_ASK_HOST;            // (the 64-bit path sums are exact for up to 2^16 nodes of 12-digit coordinates; the host refuses such a line: DESIGN §8)
    o.Tlen = v[6]; o.Ts = v[7]; o.Te = v[8];
    return 0;
}

// ---- the exact routine shared out over the lanes of a wave, node lengths computed ONCE per line ------------------------
// Per-node scratch (LDS in k_classify_slow_wave, plain arrays in tests/hostsim): what the reference recomputes for every
// link is kept per node.  phase 1: lane l takes the nodes l (mod nlanes): strand of the name (str.split quirk), node id,
// get_node_len or the exception it raises.  phase 2 (after a barrier): lane l takes the links l (mod nlanes).
// stride: node j's entries sit at index j * stride (1: one line per array; 64: the lanes of a wave interleaved, one line per lane)
struct NodeScratch { SVJG_TAB_AS uint32_t *id; SVJG_TAB_AS int64_t *len; SVJG_TAB_AS uint8_t *rc; SVJG_TAB_AS uint8_t *strand; uint32_t cap; uint32_t stride = 1; };

// can a node's name be taken by its id where its strand is asked for?  (a node of the graph whose name stands inside no other node's name:
// slow_wave_strands has the argument)
SVJG_HD bool slow_node_clean(const GraphView &g, uint32_t id) { return id != NONE32 && !(g.nodes[id].row >> 31); }

// pieces: table of the path's pieces (strand_of_pieces; colons, keys: where each has its ':' and what stands around it, optional), or
// nullptr: the path is searched byte by byte
template <class P>
SVJG_HD int slow_wave_phase1(const GraphView &g, P t, const SlowLine &ln, NodeScratch &ns, uint32_t lane, uint32_t nlanes, uint64_t *order,
                             const SVJG_TAB_AS uint32_t *pieces = nullptr, const SVJG_TAB_AS uint16_t *colons = nullptr, const SVJG_TAB_AS uint64_t *keys = nullptr) {
    NameRef nm{0, 0}; uint64_t pos = ln.ps; bool more = true;
    if (pieces) {                                                    // node j is piece j (an unoriented path's node: the piece without its last byte)
        for (uint32_t j = lane; j < ln.k; j += nlanes) {
            nm.s = pieces[j] & 0xFFFFu; nm.e = nm.s + (pieces[j] >> 16) - (ln.oriented ? 0u : 1u);
            uint32_t st = 0;
            int rc = strand_of_pieces(t, ln.ps, pieces, colons, keys, j, nm, st);
            if (rc) { *order = (1ull << 32) | j; return rc; }
            int64_t l1 = 0;
            ns.strand[j * ns.stride] = (uint8_t)st;
            bool alt;
            int64_t tl;
            const uint32_t id = resolve_name(g, t, nm, &alt, &tl); // (once: the id for the links, the form and the record's length for get_node_len)
            ns.id[j * ns.stride] = id;
            ns.rc[j * ns.stride] = (uint8_t)node_len_resolved(g, t, nm, id, alt, l1, tl);
            ns.len[j * ns.stride] = l1;
        }
        return 0;
    }
    if (nlanes == 1u && ln.oriented) {
        // one lane, the whole line (k_classify_slow; r05): every node is resolved in ONE walk, which also notes the node's own orientation
        // mark; if the line is clean (every name a graph node that stands inside no other node's name: slow_wave_strands has the
        // argument) a node's strand is the mark of the first node with the same id — filled in place, front to back — and the search
        // through the path's text is not needed; else a second walk searches as before.
        bool clean = true;
        uint32_t j = 0;
        for (; next_node(t, ln.pe, true, pos, nm); ++j) {
            int64_t l1 = 0;
            bool alt;
            int64_t tl;
            const uint32_t id = resolve_name(g, t, nm, &alt, &tl);
            ns.id[j * ns.stride] = id;
            ns.rc[j * ns.stride] = (uint8_t)node_len_resolved(g, t, nm, id, alt, l1, tl);
            ns.len[j * ns.stride] = l1;
            ns.strand[j * ns.stride] = t[nm.s - 1] == '>' ? 0u : 1u;
            clean = clean && slow_node_clean(g, id);
        }
        if (clean) {
            for (uint32_t i = 1; i < j; ++i) {
                const uint32_t x = ns.id[i * ns.stride];
                uint32_t f = 0;
                while (ns.id[f * ns.stride] != x) ++f;
                if (f != i) ns.strand[i * ns.stride] = ns.strand[f * ns.stride];
            }
            return 0;
        }
        pos = ln.ps;
        for (uint32_t i = 0; next_node(t, ln.pe, true, pos, nm); ++i) {
            uint32_t st = 0;
            int rc = strand_of(t, ln.ps, ln.pe, nm, st);
            if (rc) { *order = (1ull << 32) | i; return rc; }
            ns.strand[i * ns.stride] = (uint8_t)st;
        }
        return 0;
    }
    for (uint32_t j = 0; j < lane && more; ++j) more = next_node(t, ln.pe, ln.oriented, pos, nm);
    for (uint32_t j = lane; more && next_node(t, ln.pe, ln.oriented, pos, nm); j += nlanes) {
        uint32_t st = 0;
        int rc = strand_of(t, ln.ps, ln.pe, nm, st);                 // get_aln_links walks every node first (:203-209)
        if (rc) { *order = (1ull << 32) | j; return rc; }
        int64_t l1 = 0;
        ns.strand[j * ns.stride] = (uint8_t)st;
        bool alt;
        int64_t tl;
        const uint32_t id = resolve_name(g, t, nm, &alt, &tl);     // (once: the id for the links, the form and the record's length for get_node_len)
        ns.id[j * ns.stride] = id;
        ns.rc[j * ns.stride] = (uint8_t)node_len_resolved(g, t, nm, id, alt, l1, tl);
        ns.len[j * ns.stride] = l1;
        NameRef skip{0, 0};
        for (uint32_t q = 1; q < nlanes && more; ++q) more = next_node(t, ln.pe, ln.oriented, pos, skip);
    }
    return 0;
}

// Phase 1 in two steps (r05, the one-wave-per-line kernel): every node is resolved FIRST (slow_wave_resolve: id, get_node_len or the
// exception it raises), then the strands (slow_wave_strands).  When the line is `clean` — an oriented path all of whose names are nodes of
// the graph that stand inside no other node's name (no hazard flag: svjg/graph.py: _hazards) — the search over the path's pieces is not
// needed: such a name occurs in the path's text only as a whole node (a name holds no separator, so an occurrence lies inside ONE piece,
// i.e. inside one node's name, which it then equals), so its first occurrence is the first node with the same id and the char in front of
// it that node's orientation mark (filter-alignments.py:206).  A line with ANY other name keeps the search (that name may hold a node's
// name).  A path of 200 nodes: 45 % of the kernel's time was this search.
template <class P>
SVJG_HD void slow_wave_resolve(const GraphView &g, P t, const SlowLine &ln, NodeScratch &ns, uint32_t lane, uint32_t nlanes, const SVJG_TAB_AS uint32_t *pieces) {
    for (uint32_t j = lane; j < ln.k; j += nlanes) {
        NameRef nm;
        nm.s = pieces[j] & 0xFFFFu; nm.e = nm.s + (pieces[j] >> 16) - (ln.oriented ? 0u : 1u);
        int64_t l1 = 0;
        bool alt;
        int64_t tl;
        const uint32_t id = resolve_name(g, t, nm, &alt, &tl);
        ns.id[j * ns.stride] = id;
        ns.rc[j * ns.stride] = (uint8_t)node_len_resolved(g, t, nm, id, alt, l1, tl);
        ns.len[j * ns.stride] = l1;
    }
}
template <class P>
SVJG_HD int slow_wave_strands(P t, const SlowLine &ln, NodeScratch &ns, uint32_t lane, uint32_t nlanes, uint64_t *order, const SVJG_TAB_AS uint32_t *pieces,
                              const SVJG_TAB_AS uint16_t *colons, const SVJG_TAB_AS uint64_t *keys, bool clean, bool oneway) {
    for (uint32_t j = lane; j < ln.k; j += nlanes) {
        uint32_t st = 0;
        if (clean) {
            uint32_t f = j;
            if (!oneway) { const uint32_t x = ns.id[j * ns.stride]; f = 0; while (ns.id[f * ns.stride] != x) ++f; }
            st = t[(pieces[f] & 0xFFFFu) - 1u] == '>' ? 0u : 1u;       // (an oriented path: a mark stands in front of every piece)
        } else {
            NameRef nm;
            nm.s = pieces[j] & 0xFFFFu; nm.e = nm.s + (pieces[j] >> 16) - (ln.oriented ? 0u : 1u);
            const int rc = strand_of_pieces(t, ln.ps, pieces, colons, keys, j, nm, st);
            if (rc) { *order = (1ull << 32) | j; return rc; }
        }
        ns.strand[j * ns.stride] = (uint8_t)st;
    }
    return 0;
}

template <class Emit>
SVJG_HD int slow_wave_phase2(const GraphView &g, const SlowLine &ln, const NodeScratch &ns, Emit &emit, uint32_t lane, uint32_t nlanes, uint64_t *order) {
    const uint32_t S = ns.stride;
    for (uint32_t i = lane; i + 1 < ln.k; i += nlanes) {
        const uint32_t lid = ns.id[i * S], rid = ns.id[(i + 1) * S];
        if (lid == NONE32 || rid == NONE32) continue;
        const uint32_t ei = edge_find(g, lid, ns.strand[i * S], rid, ns.strand[(i + 1) * S]);
        if (ei == NONE32) continue;
        const svjg_edge ed = g.edges[ei];
        const uint32_t nh = ed.meta >> 2;
        if (!nh) continue;
        // list.index of both names (:269-271): a name that resolves has one spelling, so equal names <=> equal ids
        uint32_t il = 0, ir = 0;
        while (ns.id[il * S] != lid) ++il;
        while (ns.id[ir * S] != rid) ++ir;
        int64_t left = 0, right = 0;
        for (uint32_t j = 0; j <= il; ++j) { if (ns.rc[j * S]) { *order = (2ull << 32) | i; return ns.rc[j * S]; } left += ns.len[j * S]; }
        if (g.dover_list) { *order = (2ull << 32) | i; return SVJG_EXC_TYPE_ERROR; }   // int >= list (:269), before the right sum is formed
        for (uint32_t j = ir; j < ln.k; ++j) { if (ns.rc[j * S]) { *order = (2ull << 32) | i; return ns.rc[j * S]; } right += ns.len[j * S]; }
        if (left - ln.Ts >= (int64_t)g.d_over && right - (ln.Tlen - ln.Te - 1) >= (int64_t)g.d_over)
            for (uint32_t j = 0; j < nh; ++j) { uint32_t hv = edge_hit(g, ed, j); emit(hv >> 1, hv & 1u); }
    }
    return 0;
}

// One line, content t[s, e).  emit(slot, allele) is called once per appended alignment text.
// Returns 0 or the SVJG_EXC_* class the reference would die with.
// lane / nlanes: several lanes may share one line: everyone runs the per-line part (same result everywhere), lane l takes the
// path nodes j = l (mod nlanes) of the strand walk and the links i = l (mod nlanes); *order then tells where in the
// reference's sequence of steps the returned error sits, so that the caller can keep the one the reference meets first.
template <class P, class Emit>
SVJG_HD int slow_line(const GraphView &g, P t, uint64_t s, uint64_t e, Emit &emit, uint32_t lane = 0, uint32_t nlanes = 1, uint64_t *order = nullptr) {
    if (order) *order = 0;
    SlowLine ln;
    { int rc = slow_prologue(t, s, e, ln); if (rc) return rc; }
    const uint64_t ps = ln.ps, pe = ln.pe; const bool oriented = ln.oriented; const uint32_t k = ln.k;
    if (k < 2) return 0;
    // (cooperating lanes must do their shares in the SAME loop iterations, or a wave would run them one after the other:
    //  lane l walks to node l first and then advances nlanes nodes per iteration)
    { NameRef nm{0, 0}; uint64_t pos = ps; bool more = true;       // get_aln_links walks every node first
      for (uint32_t j = 0; j < lane && more; ++j) more = next_node(t, pe, oriented, pos, nm);
      for (uint32_t j = lane; more && next_node(t, pe, oriented, pos, nm); j += nlanes) {
          uint32_t st = 0; int rc = strand_of(t, ps, pe, nm, st);
          if (rc) { if (order) *order = (1ull << 32) | j; return rc; }
          NameRef skip{0, 0};
          for (uint32_t q = 1; q < nlanes && more; ++q) more = next_node(t, pe, oriented, pos, skip);
      } }
    const int64_t Tlen = ln.Tlen, Ts = ln.Ts, Te = ln.Te;
    NameRef L{0, 0}, R{0, 0};
    uint64_t posL = ps;
    bool more = true;
    for (uint32_t j = 0; j <= lane && more; ++j) more = next_node(t, pe, oriented, posL, L);      // L = node number `lane`
    for (uint32_t i = lane; more && i + 1 < k; i += nlanes) {
        uint32_t sl = 0, sr = 0;
        next_node(t, pe, oriented, posL, R);                       // R = node i + 1
        NameRef Lcur = L;
        // the lane's next link starts nlanes nodes further on (nlanes == 1: at R)
        L = R;
        for (uint32_t q = 1; q < nlanes && more; ++q) more = next_node(t, pe, oriented, posL, L);
        if (order) *order = (2ull << 32) | i;                      // (where an error of this link sits)
        strand_of(t, ps, pe, Lcur, sl);
        strand_of(t, ps, pe, R, sr);
        uint32_t lid = resolve_name(g, t, Lcur, nullptr), rid = resolve_name(g, t, R, nullptr);
        if (lid == NONE32 || rid == NONE32) continue;
        uint32_t ei = edge_find(g, lid, sl, rid, sr);
        if (ei == NONE32) continue;
        svjg_edge ed = g.edges[ei];
        uint32_t nh = ed.meta >> 2;
        if (!nh) continue;
        // list.index of both names (:269-271), then the node lengths up to / from there, in the reference's order
        uint32_t il = 0, ir = 0; NameRef nm{0, 0}; uint64_t pos = ps;
        for (;; ++il) { next_node(t, pe, oriented, pos, nm); if (nm.e - nm.s == Lcur.e - Lcur.s && bytes_eq(t, nm.s, Lcur.s, Lcur.e - Lcur.s)) break; }
        pos = ps;
        for (;; ++ir) { next_node(t, pe, oriented, pos, nm); if (nm.e - nm.s == R.e - R.s && bytes_eq(t, nm.s, R.s, R.e - R.s)) break; }
        int64_t left = 0, right = 0, l1;
        pos = ps;
        for (uint32_t j = 0; j <= il; ++j) { next_node(t, pe, oriented, pos, nm); int rc = generic_node_len(g, t, nm, l1); if (rc) return rc; left += l1; }
        if (g.dover_list) return SVJG_EXC_TYPE_ERROR;             // int >= list (:269), before the right sum is formed
        pos = ps;
        for (uint32_t j = 0; j < k; ++j) {
            next_node(t, pe, oriented, pos, nm);
            if (j < ir) continue;
            int rc = generic_node_len(g, t, nm, l1); if (rc) return rc; right += l1;
        }
        if (left - Ts >= (int64_t)g.d_over && right - (Tlen - Te - 1) >= (int64_t)g.d_over)
            for (uint32_t j = 0; j < nh; ++j) { uint32_t hv = edge_hit(g, ed, j); emit(hv >> 1, hv & 1u); }
    }
    return 0;
}

}  // namespace svjg
