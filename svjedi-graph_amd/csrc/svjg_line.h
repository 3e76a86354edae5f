// Per-alignment classification routines of the HIP path (one GAF line per lane).
//
// Two tiers, both run on the GPU:
//   fast_line  - streaming parse of a line staged in LDS.  Handles the regular case (canonical node names
//                that exist in the graph, no repeated node, no name that is a substring of another, path
//                length column consistent with the node names, plain decimal columns).  Anything else is
//                DEFERRED, never guessed.
//   slow_line  - exact string-level evaluation straight from HBM for the deferred lines, following the
//                reference's semantics to the letter, including the exception it would raise.
//
// What they replace in the reference (/root/reference/filter-alignments.py):
//   read_gaf_line :184-198, extract_nodes :351-373, get_aln_links :200-219, reverse_link :221-225,
//   the d_link_sv probes :141-153, check_bkpt_overlap :258-273, get_node_start/end/len :328-349.
//
// The functions are plain C++ on raw pointers so that tests/hostsim can also compile them with g++ and
// run them against the oracle without a GPU (test harness only; the shipped library has no CPU path).
#pragma once
#include <stdint.h>
#include "../../include/svjg.h"

#ifndef SVJG_HD
#define SVJG_HD __host__ __device__ inline
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
};

struct Pending {                 // a hit whose breakpoint-overlap test is still open
    uint32_t hit;                // slot << 1 | allele       (after fast_line: slot)
    uint32_t pre;                // path length through the link's left node   (after fast_line: n_ref | n_alt << 16)
};

enum { LINE_OK = 0, LINE_DEFER = -1 };   // fast_line returns LINE_OK or a negative defer-site number (diagnostics)

SVJG_HD bool py_space(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }

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
// fast path
// ---------------------------------------------------------------------------------------------------

// plain unsigned decimal column followed by a tab (or the end of the line when `last_ok`)
template <class P>
SVJG_HD bool col_uint(P t, uint32_t &p, uint32_t e, uint64_t &v, bool last_ok) {
    uint32_t nd = 0;
    v = 0;
    while (p < e) {
        uint32_t d = (uint32_t)t[p] - '0';
        if (d > 9) break;
        v = v * 10 + d;
        ++nd; ++p;
    }
    if (nd == 0 || nd > 18) return false;
    if (p < e) { if (t[p] != '\t') return false; ++p; return true; }
    return last_ok;
}

template <class P>
SVJG_HD bool col_skip(P t, uint32_t &p, uint32_t e) {
    while (p < e && t[p] != '\t') ++p;
    if (p >= e) return false;
    ++p;
    return true;
}

// One line, text in [s, e) of `t` (terminator excluded).  On LINE_OK, out[0..*n_out) holds the line's
// informative SVs as (hit = slot, pre = n_ref | n_alt << 16).
//
// Order of work: columns 1-5, then a light scan to the end of the path column so that Tlen/Ts/Te (columns
// 7-9) are known before the path is walked; every link's overlap test is then decided on the spot and the
// per-lane list only ever holds final, merged entries.
template <class P, class O>
SVJG_HD int fast_line(const GraphView &g, P t, uint32_t s, uint32_t e, O out, uint32_t out_cap, uint32_t *n_out) {
    *n_out = 0;
    while (e > s && py_space(t[e - 1])) --e;
    uint32_t p = s;
    uint64_t tmp;
    if (!col_skip(t, p, e)) return -1;
    if (!col_uint(t, p, e, tmp, false) || !col_uint(t, p, e, tmp, false) || !col_uint(t, p, e, tmp, false)) return -2;
    if (!col_skip(t, p, e)) return -3;
    uint32_t ps = p;
    if (!col_skip(t, p, e)) return -14;
    uint32_t pe = p - 1;                                                      // the tab that ends the path column
    uint64_t Tlen, Ts, Te, am, alen, aq;
    if (!col_uint(t, p, e, Tlen, false) || !col_uint(t, p, e, Ts, false) || !col_uint(t, p, e, Te, false)) return -15;
    if (!col_uint(t, p, e, am, false) || !col_uint(t, p, e, alen, false) || !col_uint(t, p, e, aq, true)) return -16;
    if (alen == 0) return -17;                                                // ZeroDivisionError unless an id:f: tag exists
    // a link passes iff  pre_L - Ts >= d_over  and  (Tlen - pre_L) - (Tlen - Te - 1) >= d_over   (pre_L = path
    // length through its left node; needs sum(node lengths) == Tlen, verified below)
    const uint64_t lo_ok = Ts + g.d_over;

    uint64_t pre = 0, seen1 = 0, seen2 = 0;
    uint32_t m = 0, k = 0, prev_id = NONE32, prev_or = 0;
    uint32_t prev_cidx = NONE32, prev_ch = 0, prev_cl = 0;
    p = ps;
    while (p < pe && (t[p] == '>' || t[p] == '<')) {
        uint32_t orient = t[p] == '<';
        ++p;
        uint32_t ns = p, colon = NONE32, nd1 = 0, nd2 = 0, sep = 0, bad = 0, h = FNV_INIT, hc = 0;
        uint64_t v1 = 0, v2 = 0;
        while (p < pe) {
            uint32_t c = t[p];
            if (c == '>' || c == '<') break;
            if (c == ':') { colon = p; hc = h; v1 = v2 = 0; nd1 = nd2 = sep = bad = 0; }
            else if (colon != NONE32) {
                uint32_t d = c - '0';
                if (d <= 9) {
                    if (!sep) { bad |= (nd1 == 1 && v1 == 0); v1 = v1 * 10 + d; ++nd1; }
                    else      { bad |= (nd2 == 1 && v2 == 0); v2 = v2 * 10 + d; ++nd2; }
                } else if ((c == '-' || c == '.') && !sep && nd1) sep = c;
                else bad = 1;
            }
            h = (h ^ c) * FNV_PRIME;
            ++p;
        }
        if (colon == NONE32 || !sep || !nd1 || !nd2 || bad || nd1 > 10 || nd2 > 10 || v1 > 0xFFFFFFFFull || v2 > 0xFFFFFFFFull)
            return -4;
        uint32_t clen = colon - ns, cidx;
        if (prev_cidx != NONE32 && hc == prev_ch && clen == prev_cl) {
            uint32_t o = g.chrom_off[prev_cidx], j = 0;
            while (j < clen && g.chrom_names[o + j] == t[ns + j]) ++j;
            cidx = (j == clen) ? prev_cidx : chrom_lookup(g, t, ns, clen, hc);
        } else cidx = chrom_lookup(g, t, ns, clen, hc);
        if (cidx == NONE32) return -5;
        prev_cidx = cidx; prev_ch = hc; prev_cl = clen;
        uint32_t kind = sep == '.';
        if (kind && v2 >= 32768) return -6;
        uint64_t key = node_key(cidx, (uint32_t)v1, kind, kind ? (uint32_t)v2 : 0);
        uint32_t id = NONE32;
        if (prev_id != NONE32) {
            if (g.nodes[prev_id + 1].key == key) id = prev_id + 1;            // sentinel key never matches
            else if (prev_id > 0 && g.nodes[prev_id - 1].key == key) id = prev_id - 1;
            else if (prev_id + 2 <= g.n_nodes && g.nodes[prev_id + 2].key == key) id = prev_id + 2;
            else if (prev_id > 1 && g.nodes[prev_id - 2].key == key) id = prev_id - 2;
        }
        if (id == NONE32) id = node_search(g, cidx, key);
        if (id == NONE32) return -7;
        svjg_node nd = g.nodes[id];
        if (nd.row & 0x80000000u) return -8;
        uint64_t len;
        if (kind) { if (nd.aux == SVJG_LEN_UNKNOWN) return -9; len = nd.aux; }
        else { if (nd.aux != (uint32_t)v2) return -10; len = v2 - v1 + 1; }
        // two-hash filter for "this node was already on the path" (exactness is the deferred path's job)
        uint64_t b1 = 1ull << (id & 63), b2 = 1ull << ((id * 0x9E3779B1u) >> 26);
        if ((seen1 & b1) && (seen2 & b2)) return -11;
        seen1 |= b1; seen2 |= b2;
        if (k && pre >= lo_ok && pre + (g.d_over - 1) <= Te) {
            uint32_t ei = edge_find(g, prev_id, prev_or, id, orient);
            if (ei != NONE32) {
                svjg_edge ed = g.edges[ei];
                uint32_t nh = ed.meta >> 2;
                for (uint32_t j = 0; j < nh; ++j) {
                    uint32_t hv = edge_hit(g, ed, j), slot = hv >> 1, add = (hv & 1) ? 0x10000u : 1u, q = 0;
                    while (q < m && out[q].hit != slot) ++q;
                    if (q == m) {
                        if (m == out_cap) return -12;
                        out[m].hit = slot; out[m].pre = add; ++m;
                    } else {
                        out[q].pre += add;
                        if ((out[q].pre & 0xFFFFu) == 0xFFFFu || (out[q].pre >> 16) == 0xFFFFu) return -12;
                    }
                }
            }
        }
        pre += len;
        if (pre > 0xFFFFFFFFull) return -13;
        prev_id = id; prev_or = orient; ++k;
    }
    if (k == 0 || p != pe) return -14;
    if (k < 2) return LINE_OK;
    if (pre != Tlen) return -18;
    *n_out = m;
    return LINE_OK;
}

// ---------------------------------------------------------------------------------------------------
// exact path
// ---------------------------------------------------------------------------------------------------

constexpr int64_t BIGV = (int64_t)1 << 61;

// Python int(): blanks, sign, digits with single inner underscores (ASCII subset)
template <class P>
SVJG_HD bool py_int(P t, uint64_t a, uint64_t b, int64_t &out) {
    while (a < b && py_space(t[a])) ++a;
    while (b > a && py_space(t[b - 1])) --b;
    bool neg = false;
    if (a < b && (t[a] == '+' || t[a] == '-')) { neg = t[a] == '-'; ++a; }
    if (a >= b || (uint32_t)t[a] - '0' > 9) return false;
    int64_t v = 0; bool us = false;
    for (; a < b; ++a) {
        uint32_t c = t[a];
        if (c == '_') { if (us) return false; us = true; continue; }
        if (c - '0' > 9) return false;
        us = false;
        if (v < BIGV) v = v * 10 + (int64_t)(c - '0');
    }
    if (us) return false;
    if (v > BIGV) v = BIGV;
    out = neg ? -v : v;
    return true;
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
SVJG_HD bool py_float_ok(P t, uint64_t a, uint64_t b) {
    while (a < b && py_space(t[a])) ++a;
    while (b > a && py_space(t[b - 1])) --b;
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

struct NameRef { uint64_t s, e; };     // node name = t[s, e)

// i-th node name of the path t[ps, pe).  oriented: non-empty pieces between '<' / '>' ; otherwise the
// comma-separated pieces minus their last character (filter-alignments.py:366-371).
template <class P>
SVJG_HD bool nth_node(P t, uint64_t ps, uint64_t pe, bool oriented, uint32_t i, NameRef &out) {
    uint64_t st = ps; uint32_t seen = 0;
    for (uint64_t q = ps; q <= pe; ++q) {
        bool brk = q == pe || (oriented ? (t[q] == '<' || t[q] == '>') : t[q] == ',');
        if (!brk) continue;
        if (q > st) {
            if (seen == i) { out.s = st; out.e = oriented ? q : q - 1; return true; }
            ++seen;
        }
        st = q + 1;
    }
    return false;
}

// exact name -> node id (only canonical spellings can be in the table)
template <class P>
SVJG_HD uint32_t resolve_name(const GraphView &g, P t, NameRef nm, bool *is_alt_form) {
    uint64_t colon = nm.e;
    for (uint64_t q = nm.e; q > nm.s; --q) if (t[q - 1] == ':') { colon = q - 1; break; }
    if (is_alt_form) {
        *is_alt_form = false;
        for (uint64_t q = (colon == nm.e ? nm.s : colon + 1); q < nm.e; ++q) if (t[q] == '.') *is_alt_form = true;
    }
    if (colon == nm.e) return NONE32;
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

// get_node_len (filter-alignments.py:343-349): 0 = ok, else the exception class
template <class P>
SVJG_HD int generic_node_len(const GraphView &g, P t, NameRef nm, int64_t &len) {
    bool alt;
    uint32_t id = resolve_name(g, t, nm, &alt);
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
    if (!py_int(t, d1 + 1, d2, b)) return SVJG_EXC_VALUE_ERROR;
    if (!py_int(t, c0, d1, a)) return SVJG_EXC_VALUE_ERROR;
    len = b - a + 1;
    return 0;
}

// char before the first occurrence of the name as a substring of the path (filter-alignments.py:206)
template <class P>
SVJG_HD int strand_of(P t, uint64_t ps, uint64_t pe, NameRef nm, uint32_t &strand) {
    uint64_t n = nm.e - nm.s;
    if (n == 0) return SVJG_EXC_VALUE_ERROR;                      // str.split("")
    for (uint64_t q = ps; q + n <= pe; ++q)
        if (t[q] == t[nm.s] && bytes_eq(t, q, nm.s, n)) {
            if (q == ps) return SVJG_EXC_INDEX_ERROR;             // ""[-1]
            strand = t[q - 1] == '>' ? 0u : 1u;
            return 0;
        }
    return SVJG_EXC_INDEX_ERROR;                                  // unreachable: the name is part of the path
}

// One line, content t[s, e).  emit(slot, allele) is called once per appended alignment text.
// Returns 0 or the SVJG_EXC_* class the reference would die with.
template <class P, class Emit>
SVJG_HD int slow_line(const GraphView &g, P t, uint64_t s, uint64_t e, Emit &emit) {
    while (e > s && py_space(t[e - 1])) --e;
    uint64_t fs[12], fe[12]; uint32_t nf = 0;
    { uint64_t st = s;
      for (uint64_t q = s; q <= e && nf < 12; ++q)
          if (q == e || t[q] == '\t') { fs[nf] = st; fe[nf] = q; ++nf; st = q + 1; } }
    if (nf < 12) return SVJG_EXC_VALUE_ERROR;
    int64_t v[12];
    const int cols[9] = {1, 2, 3, 6, 7, 8, 9, 10, 11};
    for (int j = 0; j < 9; ++j) if (!py_int(t, fs[cols[j]], fe[cols[j]], v[cols[j]])) return SVJG_EXC_VALUE_ERROR;
    { uint64_t last = e;                                           // "id:f:" in line  (:193-196)
      for (uint64_t q = s; q + 5 <= e; ++q)
          if (t[q] == 'i' && t[q + 1] == 'd' && t[q + 2] == ':' && t[q + 3] == 'f' && t[q + 4] == ':') last = q;
      if (last != e) {
          uint64_t a = last + 5, b = a;
          while (b < e && t[b] != '\t') ++b;
          if (!py_float_ok(t, a, b)) return SVJG_EXC_VALUE_ERROR;
      } else if (v[10] == 0) return SVJG_EXC_ZERO_DIVISION; }
    uint64_t ps = fs[5], pe = fe[5];
    if (pe == ps) return SVJG_EXC_INDEX_ERROR;                     // p[0]
    bool oriented = t[ps] == '<' || t[ps] == '>';
    uint32_t k = 0;
    { NameRef nm{0, 0}; while (nth_node(t, ps, pe, oriented, k, nm)) ++k; }
    if (k < 2) return 0;
    for (uint32_t i = 0; i < k; ++i) {                             // get_aln_links walks every node first
        NameRef nm{0, 0}; uint32_t st = 0;
        nth_node(t, ps, pe, oriented, i, nm);
        int rc = strand_of(t, ps, pe, nm, st);
        if (rc) return rc;
    }
    int64_t Tlen = v[6], Ts = v[7], Te = v[8];
    for (uint32_t i = 0; i + 1 < k; ++i) {
        NameRef L{0, 0}, R{0, 0}; uint32_t sl = 0, sr = 0;
        nth_node(t, ps, pe, oriented, i, L);
        nth_node(t, ps, pe, oriented, i + 1, R);
        strand_of(t, ps, pe, L, sl);
        strand_of(t, ps, pe, R, sr);
        uint32_t lid = resolve_name(g, t, L, nullptr), rid = resolve_name(g, t, R, nullptr);
        if (lid == NONE32 || rid == NONE32) continue;
        uint32_t ei = edge_find(g, lid, sl, rid, sr);
        if (ei == NONE32) continue;
        svjg_edge ed = g.edges[ei];
        uint32_t nh = ed.meta >> 2;
        if (!nh) continue;
        uint32_t il = 0, ir = 0; NameRef nm{0, 0};
        for (;; ++il) { nth_node(t, ps, pe, oriented, il, nm); if (nm.e - nm.s == L.e - L.s && bytes_eq(t, nm.s, L.s, L.e - L.s)) break; }
        for (;; ++ir) { nth_node(t, ps, pe, oriented, ir, nm); if (nm.e - nm.s == R.e - R.s && bytes_eq(t, nm.s, R.s, R.e - R.s)) break; }
        int64_t left = 0, right = 0, l1;
        for (uint32_t j = 0; j <= il; ++j) { nth_node(t, ps, pe, oriented, j, nm); int rc = generic_node_len(g, t, nm, l1); if (rc) return rc; left += l1; }
        for (uint32_t j = ir; j < k; ++j) { nth_node(t, ps, pe, oriented, j, nm); int rc = generic_node_len(g, t, nm, l1); if (rc) return rc; right += l1; }
        if (left - Ts >= (int64_t)g.d_over && right - (Tlen - Te - 1) >= (int64_t)g.d_over)
            for (uint32_t j = 0; j < nh; ++j) { uint32_t hv = edge_hit(g, ed, j); emit(hv >> 1, hv & 1u); }
    }
    return 0;
}

}  // namespace svjg
