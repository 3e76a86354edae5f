/* Synthetic GAF writer for the benchmark / parity configurations.
 *
 * Not part of the product path and not part of the oracle: it only manufactures inputs.
 * Random walks over the variation graph built by tools/synth.py; every line is a pure function
 * of (seed, line index) through a counter-based splitmix64 stream, so any shard of the file can be
 * regenerated independently and in parallel.
 *
 * build: gcc -O2 -shared -fPIC -o tools/_build/libsvjg_synth.so tools/svjg_synth.c
 */
#include <stdint.h>
#include <string.h>
#include <stdio.h>

typedef struct { uint64_t s; } rng_t;

static inline uint64_t sm64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t below(rng_t *r, uint64_t n) { return n ? sm64(&r->s) % n : 0; }

static inline char *put_u64(char *p, uint64_t v) {
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

#define KMAX 16

/* arcs: CSR over state = node*2 + orient (orient 0 = '+', 1 = '-'); arc_to = next state;
 * arc_sv = index of the SV whose ALT allele this arc is, or -1 for a reference arc. */
long svjg_synth_gaf(const char *names, const uint32_t *name_off, const uint32_t *node_len,
                    uint32_t n_ref_nodes, const uint32_t *arc_ptr, const uint32_t *arc_to,
                    const int32_t *arc_sv, const uint8_t *sv_gt, uint64_t seed,
                    uint64_t first, uint64_t n, char *out, uint64_t cap)
{
    char *p = out, *end = out + cap;
    for (uint64_t li = first; li < first + n; ++li) {
        rng_t r; r.s = seed * 0xD1342543DE82EF95ull + li * 0x2545F4914F6CDD1Dull + 1;
        sm64(&r.s);
        uint32_t st[KMAX + 2]; int k = 1;
        st[0] = (uint32_t)below(&r, n_ref_nodes) * 2u;
        int want = 1;
        while (want < 12 && below(&r, 5) != 0) ++want;           /* 1 + Geometric(p = 0.2), clipped */
        int hap = (int)(sm64(&r.s) & 1);
        while (k < want) {
            uint32_t cur = st[k - 1], a0 = arc_ptr[cur], a1 = arc_ptr[cur + 1];
            int32_t pick = -1, refarc = -1;
            for (uint32_t a = a0; a < a1; ++a) {
                if (arc_sv[a] < 0) { if (refarc < 0) refarc = (int32_t)a; continue; }
                uint8_t g = sv_gt[arc_sv[a]];
                if (pick < 0 && (g == 2 || (g == 1 && hap))) pick = (int32_t)a;
            }
            if (pick < 0) pick = refarc;
            if (pick < 0) break;
            st[k++] = arc_to[pick];
        }
        if (k >= 2 && below(&r, 1000) == 0 && k + 2 <= KMAX) {  /* 0.1 %: revisit a node pair */
            int j = (int)below(&r, (uint64_t)(k - 1));
            memmove(&st[j + 4], &st[j + 2], (size_t)(k - j - 2) * sizeof(uint32_t));
            st[j + 2] = st[j]; st[j + 3] = st[j + 1];
            k += 2;
        }
        if (sm64(&r.s) & 1) {                                   /* reverse-strand rendering */
            for (int i = 0; i < k / 2; ++i) { uint32_t t = st[i]; st[i] = st[k - 1 - i]; st[k - 1 - i] = t; }
            for (int i = 0; i < k; ++i) st[i] ^= 1u;
        }
        uint64_t tlen = 0;
        for (int i = 0; i < k; ++i) tlen += node_len[st[i] >> 1];
        uint64_t lf = node_len[st[0] >> 1], ll = node_len[st[k - 1] >> 1];
        uint64_t ts = below(&r, lf < 400 ? lf : 400);
        uint64_t te = tlen - below(&r, ll < 400 ? ll : 400);
        if (te <= ts) { ts = 0; te = tlen; }
        uint64_t alen = te - ts, am = alen - alen / 10 - below(&r, alen / 20 + 1);
        if ((uint64_t)(end - p) < 512 + (uint64_t)k * 80) return -1;
        memcpy(p, "read", 4); p += 4; p = put_u64(p, li); *p++ = '\t';
        p = put_u64(p, alen + 37); *p++ = '\t'; p = put_u64(p, 12); *p++ = '\t'; p = put_u64(p, alen + 12); *p++ = '\t';
        *p++ = '+'; *p++ = '\t';
        for (int i = 0; i < k; ++i) {
            uint32_t nd = st[i] >> 1, l = name_off[nd + 1] - name_off[nd];
            *p++ = (st[i] & 1) ? '<' : '>';
            memcpy(p, names + name_off[nd], l); p += l;
        }
        *p++ = '\t'; p = put_u64(p, tlen); *p++ = '\t'; p = put_u64(p, ts); *p++ = '\t'; p = put_u64(p, te); *p++ = '\t';
        p = put_u64(p, am); *p++ = '\t'; p = put_u64(p, alen); *p++ = '\t'; p = put_u64(p, 60);
        memcpy(p, "\ttp:A:P\tcm:i:", 13); p += 13; p = put_u64(p, am / 12 + 1);
        memcpy(p, "\ts1:i:", 6); p += 6; p = put_u64(p, am - am / 7);
        memcpy(p, "\ts2:i:", 6); p += 6; p = put_u64(p, below(&r, 200));
        memcpy(p, "\tdv:f:0.", 8); p += 8;
        { uint64_t dv = below(&r, 1500); *p++ = (char)('0' + dv / 1000); *p++ = (char)('0' + dv / 100 % 10);
          *p++ = (char)('0' + dv / 10 % 10); *p++ = (char)('0' + dv % 10); }
        *p++ = '\n';
    }
    return (long)(p - out);
}


/* Long-read shaped lines (bench.py's secondary block; tests/golden/realshape has the hand-made originals): PacBio / ONT style read
 * names, paths whose length is long-tailed — 2 + Geometric(mean ~9) for most, a uniform 65..200 nodes for 3 % of the lines — in both
 * directions, the five minigraph tags, and on a third of the lines a cg:Z: string of 20..420 operations (none where the path alone
 * fills most of a stripe).  Same counter-based stream per line as above. */
#define KLONGMAX 304
long svjg_synth_gaf_long(const char *names, const uint32_t *name_off, const uint32_t *node_len,
                         uint32_t n_ref_nodes, const uint32_t *arc_ptr, const uint32_t *arc_to,
                         const int32_t *arc_sv, const uint8_t *sv_gt, uint64_t seed,
                         uint64_t first, uint64_t n, char *out, uint64_t cap)
{
    char *p = out, *end = out + cap;
    static const char hex[] = "0123456789abcdef";
    for (uint64_t li = first; li < first + n; ++li) {
        rng_t r; r.s = seed * 0xD1342543DE82EF95ull + li * 0x2545F4914F6CDD1Dull + 7;
        sm64(&r.s);
        uint32_t st[KLONGMAX + 4]; int k = 1;
        st[0] = (uint32_t)below(&r, n_ref_nodes) * 2u;
        int want = 2;
        if (below(&r, 100) < 3) want = 65 + (int)below(&r, 136);
        else while (want < 64 && below(&r, 9) != 0) ++want;
        int hap = (int)(sm64(&r.s) & 1);
        while (k < want) {
            uint32_t cur = st[k - 1], a0 = arc_ptr[cur], a1 = arc_ptr[cur + 1];
            int32_t pick = -1, refarc = -1;
            for (uint32_t a = a0; a < a1; ++a) {
                if (arc_sv[a] < 0) { if (refarc < 0) refarc = (int32_t)a; continue; }
                uint8_t g = sv_gt[arc_sv[a]];
                if (pick < 0 && (g == 2 || (g == 1 && hap))) pick = (int32_t)a;
            }
            if (pick < 0) pick = refarc;
            if (pick < 0) break;
            st[k++] = arc_to[pick];
        }
        if (k >= 2 && below(&r, 1000) == 0 && k + 2 <= KLONGMAX) {  /* 0.1 %: revisit a node pair */
            int j = (int)below(&r, (uint64_t)(k - 1));
            memmove(&st[j + 4], &st[j + 2], (size_t)(k - j - 2) * sizeof(uint32_t));
            st[j + 2] = st[j]; st[j + 3] = st[j + 1];
            k += 2;
        }
        if (sm64(&r.s) & 1) {
            for (int i = 0; i < k / 2; ++i) { uint32_t t = st[i]; st[i] = st[k - 1 - i]; st[k - 1 - i] = t; }
            for (int i = 0; i < k; ++i) st[i] ^= 1u;
        }
        uint64_t tlen = 0, path_bytes = 0;
        for (int i = 0; i < k; ++i) { tlen += node_len[st[i] >> 1]; path_bytes += 1 + name_off[(st[i] >> 1) + 1] - name_off[st[i] >> 1]; }
        uint64_t lf = node_len[st[0] >> 1], ll = node_len[st[k - 1] >> 1];
        uint64_t ts = below(&r, lf < 400 ? lf : 400);
        uint64_t te = tlen - below(&r, ll < 400 ? ll : 400);
        if (te <= ts) { ts = 0; te = tlen; }
        uint64_t alen = te - ts, am = alen - alen / 10 - below(&r, alen / 20 + 1);
        int n_ops = (below(&r, 3) == 0 && path_bytes < 4500) ? 20 + (int)below(&r, 401) : 0;
        if ((uint64_t)(end - p) < 768 + path_bytes + (uint64_t)n_ops * 8) return -1;
        if (li & 1) {                                            /* PacBio: movie/zmw/ccs */
            memcpy(p, "m64011_190830_220126/", 21); p += 21; p = put_u64(p, 4000000 + li % 90000000); memcpy(p, "/ccs", 4); p += 4;
        } else {                                                 /* ONT: a UUID */
            uint64_t a = sm64(&r.s), b = sm64(&r.s);
            for (int i = 0; i < 32; ++i) { if (i == 8 || i == 12 || i == 16 || i == 20) *p++ = '-'; *p++ = hex[((i < 16 ? a : b) >> (4 * (i & 15))) & 15]; }
        }
        *p++ = '\t';
        p = put_u64(p, alen + 37); *p++ = '\t'; p = put_u64(p, 12); *p++ = '\t'; p = put_u64(p, alen + 12); *p++ = '\t';
        *p++ = '+'; *p++ = '\t';
        for (int i = 0; i < k; ++i) {
            uint32_t nd = st[i] >> 1, l = name_off[nd + 1] - name_off[nd];
            *p++ = (st[i] & 1) ? '<' : '>';
            memcpy(p, names + name_off[nd], l); p += l;
        }
        *p++ = '\t'; p = put_u64(p, tlen); *p++ = '\t'; p = put_u64(p, ts); *p++ = '\t'; p = put_u64(p, te); *p++ = '\t';
        p = put_u64(p, am); *p++ = '\t'; p = put_u64(p, alen); *p++ = '\t'; p = put_u64(p, 60);
        memcpy(p, "\ttp:A:P\tcm:i:", 13); p += 13; p = put_u64(p, am / 12 + 1);
        memcpy(p, "\ts1:i:", 6); p += 6; p = put_u64(p, am - am / 7);
        memcpy(p, "\ts2:i:", 6); p += 6; p = put_u64(p, below(&r, 200));
        memcpy(p, "\tdv:f:0.", 8); p += 8;
        { uint64_t dv = below(&r, 1500); *p++ = (char)('0' + dv / 1000); *p++ = (char)('0' + dv / 100 % 10);
          *p++ = (char)('0' + dv / 10 % 10); *p++ = (char)('0' + dv % 10); }
        if (n_ops) {
            memcpy(p, "\tcg:Z:", 6); p += 6;
            for (int i = 0; i < n_ops; ++i) { p = put_u64(p, 1 + below(&r, (i & 1) ? 4 : 300)); *p++ = (i & 1) ? ((i & 2) ? 'I' : 'D') : 'M'; }
        }
        *p++ = '\n';
    }
    return (long)(p - out);
}


/* Whole-genome long-read shaped lines (BASELINE configs[4]'s shape: ~12 k SVs on 24 GRCh37-length contigs, 30x of ~20 kb reads): a read
 * STARTS at a position drawn uniformly over the genome (cum_len: running sum of the reference nodes' lengths, n_ref + 1 entries) and is
 * walked for its length, so the path has as many nodes as the read crosses breakpoints — with ~240 kb between breakpoints most lines
 * are SINGLE-node paths (which filter-alignments.py:133-134 skips).  ts / te are where the read really starts and ends in its first and
 * last node, mirrored for the half of the lines rendered on the reverse strand.  ONT-style read names, the five tags of `minigraph -x lr`
 * (no cg:Z: svjedi-graph.py:104 does not pass -c).  Same counter-based stream per line as above. */
long svjg_synth_gaf_reads(const char *names, const uint32_t *name_off, const uint32_t *node_len,
                          uint32_t n_ref_nodes, const uint32_t *arc_ptr, const uint32_t *arc_to,
                          const int32_t *arc_sv, const uint8_t *sv_gt, uint64_t seed,
                          uint64_t first, uint64_t n, char *out, uint64_t cap, const uint64_t *cum_len)
{
    char *p = out, *end = out + cap;
    static const char hex[] = "0123456789abcdef";
    const uint64_t genome = cum_len[n_ref_nodes];
    for (uint64_t li = first; li < first + n; ++li) {
        rng_t r; r.s = seed * 0xD1342543DE82EF95ull + li * 0x2545F4914F6CDD1Dull + 11;
        sm64(&r.s);
        const uint64_t at = below(&r, genome);
        uint32_t lo = 0, hi = n_ref_nodes;                         /* the node that holds position `at` */
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (cum_len[mid] <= at) lo = mid; else hi = mid; }
        uint64_t want = 4000 + below(&r, 32000);                   /* read length: 4..36 kb, one in twenty up to 120 kb */
        if (below(&r, 20) == 0) want += below(&r, 84000);
        int hap = (int)(sm64(&r.s) & 1);
        uint32_t st[KLONGMAX + 4]; int k = 1;
        st[0] = lo * 2u;
        uint64_t ts = at - cum_len[lo], tlen = node_len[lo], left = want, room = node_len[lo] - ts, used_last = 0;
        for (;;) {
            if (left <= room) { used_last = (k == 1 ? ts : 0) + left; break; }
            left -= room;
            uint32_t cur = st[k - 1], a0 = arc_ptr[cur], a1 = arc_ptr[cur + 1];
            int32_t pick = -1, refarc = -1;
            for (uint32_t a = a0; a < a1; ++a) {
                if (arc_sv[a] < 0) { if (refarc < 0) refarc = (int32_t)a; continue; }
                uint8_t g = sv_gt[arc_sv[a]];
                if (pick < 0 && (g == 2 || (g == 1 && hap))) pick = (int32_t)a;
            }
            if (pick < 0) pick = refarc;
            if (pick < 0 || k >= KLONGMAX) { used_last = node_len[st[k - 1] >> 1]; break; }   /* the contig (or the table) ends: the read is clipped */
            st[k++] = arc_to[pick];
            room = node_len[st[k - 1] >> 1];
            tlen += room;
        }
        uint64_t te = tlen - (node_len[st[k - 1] >> 1] - used_last);
        if (te <= ts) { ts = 0; te = tlen; }
        if (sm64(&r.s) & 1) {                                    /* reverse-strand rendering: the path backwards, coordinates mirrored */
            for (int i = 0; i < k / 2; ++i) { uint32_t t = st[i]; st[i] = st[k - 1 - i]; st[k - 1 - i] = t; }
            for (int i = 0; i < k; ++i) st[i] ^= 1u;
            const uint64_t a = tlen - te, b = tlen - ts;
            ts = a; te = b;
        }
        uint64_t path_bytes = 0;
        for (int i = 0; i < k; ++i) path_bytes += 1 + name_off[(st[i] >> 1) + 1] - name_off[st[i] >> 1];
        uint64_t alen = te - ts, am = alen - alen / 10 - below(&r, alen / 20 + 1);
        if ((uint64_t)(end - p) < 768 + path_bytes) return -1;
        { uint64_t a = sm64(&r.s), b = sm64(&r.s);
          for (int i = 0; i < 32; ++i) { if (i == 8 || i == 12 || i == 16 || i == 20) *p++ = '-'; *p++ = hex[((i < 16 ? a : b) >> (4 * (i & 15))) & 15]; } }
        *p++ = '\t';
        p = put_u64(p, alen + 37); *p++ = '\t'; p = put_u64(p, 12); *p++ = '\t'; p = put_u64(p, alen + 12); *p++ = '\t';
        *p++ = '+'; *p++ = '\t';
        for (int i = 0; i < k; ++i) {
            uint32_t nd = st[i] >> 1, l = name_off[nd + 1] - name_off[nd];
            *p++ = (st[i] & 1) ? '<' : '>';
            memcpy(p, names + name_off[nd], l); p += l;
        }
        *p++ = '\t'; p = put_u64(p, tlen); *p++ = '\t'; p = put_u64(p, ts); *p++ = '\t'; p = put_u64(p, te); *p++ = '\t';
        p = put_u64(p, am); *p++ = '\t'; p = put_u64(p, alen); *p++ = '\t'; p = put_u64(p, 60);
        memcpy(p, "\ttp:A:P\tcm:i:", 13); p += 13; p = put_u64(p, am / 12 + 1);
        memcpy(p, "\ts1:i:", 6); p += 6; p = put_u64(p, am - am / 7);
        memcpy(p, "\ts2:i:", 6); p += 6; p = put_u64(p, below(&r, 200));
        memcpy(p, "\tdv:f:0.", 8); p += 8;
        { uint64_t dv = below(&r, 1500); *p++ = (char)('0' + dv / 1000); *p++ = (char)('0' + dv / 100 % 10);
          *p++ = (char)('0' + dv / 10 % 10); *p++ = (char)('0' + dv % 10); }
        *p++ = '\n';
    }
    return (long)(p - out);
}
