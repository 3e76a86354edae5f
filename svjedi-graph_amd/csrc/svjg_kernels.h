// HIP kernels of libsvjg_hip.so (gfx950 / MI355X, wave64).  No MFMA anywhere: this is byte / integer
// work bounded by HBM reads.
//
//   k_classify_main  one workgroup per ~44 KB stripe of GAF text:
//                      A  coalesced 16 B/lane loads HBM -> LDS (the only HBM read of the text)
//                      B  line-terminator scan over LDS, block prefix sum -> line-start list in LDS
//                      C  one line per lane: svjg::fast_line (streaming parse + node / link lookups in
//                         the L2 / Infinity-Cache resident graph tables)
//                      D  wave-level commit: packed 64-bit (ref | alt << 32) atomics into the per-SV
//                         count vector, hit records and deferred-line offsets appended with one
//                         wave-aggregated atomic each
//   k_classify_slow  one lane per deferred line, exact string path (svjg::slow_line) straight from HBM
//   k_logfact_*      log10(i!) table in double-double for the binomial term
//   k_genotype       one VCF row per lane, fp64 / double-double likelihoods (predict-genotype.py:281-325)
#pragma once
#include <hip/hip_runtime.h>
#include "svjg_line.h"

namespace svjg {

constexpr uint32_t WG = 256;
constexpr uint32_t CHUNK = 44 * 1024;            // bytes of text owned by one workgroup iteration
constexpr uint32_t LOOK = 4 * 1024;              // look-ahead so that lines starting in the chunk are complete
constexpr uint32_t TEXT = CHUNK + LOOK;          // 48 KB staged in LDS
constexpr uint32_t SPAN = TEXT / WG;             // 192 B of terminator scan per lane
constexpr uint32_t PIECES = SPAN / 16;           // 12
constexpr uint32_t MAXSTARTS = TEXT / 24 + 8;    // a valid line has >= 24 bytes incl. its terminator
constexpr uint32_t HMAX = 12;                    // informative SVs per alignment kept by the fast path
constexpr uint32_t DICT_LDS_MAX = 8 * 1024;      // chromosome dictionary is copied to LDS when it fits

// status words (device)
struct DevStatus {
    unsigned long long n_lines;
    unsigned long long n_deferred;       // entries appended to the deferred list
    unsigned long long n_recs;           // hit records appended
    unsigned long long err;              // min over (file offset << 3 | exception class); ~0 = none
    unsigned int non_ascii;
    unsigned int overflow;               // bit 0: deferred list, bit 1: hit-record buffer
};

struct ClassifyArgs {
    const uint8_t *gaf;                  // resident text, allocation padded with >= TEXT + 64 zero bytes
    uint64_t n_bytes;
    uint64_t base_offset;
    GraphView g;                         // global-memory views
    uint32_t dict_names_len;             // bytes of chromosome names
    uint32_t dict_in_lds;
    uint32_t all_slow;
    uint32_t want_hits;
    uint32_t n_chunks;
    unsigned long long *counts;          // [n_slots] ref | alt << 32
    uint64_t *deferred;  uint64_t deferred_cap;
    svjg_hitrec *recs;   uint64_t rec_cap;
    DevStatus *st;
};

struct LaneList {                        // per-lane list laid out [entry][lane]: conflict-free ds_write_b64
    Pending *base;
    __device__ Pending &operator[](uint32_t j) const { return base[j * WG]; }
};

__device__ inline uint32_t eq_mask4(uint32_t x, uint32_t pat) {        // 4-bit mask of bytes equal to pat's byte
    uint32_t t = x ^ pat;
    uint32_t m = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);   // 0x80 where the byte is zero
    return (((m >> 7) * 0x00204081u) >> 21) & 0xFu;
}

__device__ inline uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
    uint32_t lane = __lane_id();
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}

__global__ __launch_bounds__(WG) void k_classify_main(ClassifyArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *text = lds;                                               // TEXT + 16
    uint16_t *starts = (uint16_t *)(lds + TEXT + 16);                  // MAXSTARTS + 8 (u16)
    Pending *lists = (Pending *)(lds + TEXT + 16 + ((MAXSTARTS + 8) * 2 + 15) / 16 * 16);   // HMAX * WG
    uint32_t *misc = (uint32_t *)((uint8_t *)lists + HMAX * WG * sizeof(Pending));           // 16 words
    uint8_t *dict = (uint8_t *)(misc + 16);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    GraphView g = a.g;
    if (a.dict_in_lds) {
        // names | off[n+1] | lo[n+1] | hash[mask+1]   (names padded to 4 B)
        uint32_t nb = (a.dict_names_len + 3) & ~3u, n1 = g.n_chrom + 1, hs = g.hash_mask + 1;
        uint32_t *d_off = (uint32_t *)(dict + nb), *d_lo = d_off + n1, *d_hash = d_lo + n1;
        for (uint32_t i = tid; i < a.dict_names_len; i += WG) dict[i] = g.chrom_names[i];
        for (uint32_t i = tid; i < n1; i += WG) { d_off[i] = g.chrom_off[i]; d_lo[i] = g.chrom_lo[i]; }
        for (uint32_t i = tid; i < hs; i += WG) d_hash[i] = g.chrom_hash[i];
        g.chrom_names = dict; g.chrom_off = d_off; g.chrom_lo = d_lo; g.chrom_hash = d_hash;
    }

    const uint64_t padded = (a.n_bytes + 15) & ~15ull;
    unsigned long long wg_lines = 0;

    for (uint32_t chunk = blockIdx.x; chunk < a.n_chunks; chunk += gridDim.x) {
        const uint64_t c0 = (uint64_t)chunk * CHUNK;
        const uint32_t V = (uint32_t)((a.n_bytes - c0 < (uint64_t)TEXT) ? (a.n_bytes - c0) : (uint64_t)TEXT);   // valid bytes staged

        // ---- A: HBM -> LDS ---------------------------------------------------------------------
        uint32_t hi_bits = 0;
#pragma unroll 4
        for (uint32_t i = tid; i < TEXT / 16; i += WG) {
            uint64_t off = c0 + (uint64_t)i * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (off < padded) v = *(const uint4 *)(a.gaf + off);
            hi_bits |= v.x | v.y | v.z | v.w;
            *(uint4 *)(text + i * 16) = v;
        }
        if (tid == 0) { misc[0] = 0; misc[1] = 0; }
        if (hi_bits & 0x80808080u) a.st->non_ascii = 1;
        __syncthreads();

        // ---- B: terminators -> line starts ----------------------------------------------------------
        // terminator = '\n', or a '\r' not followed by '\n' (Python universal newlines)
        uint32_t mask[PIECES / 2];                                       // two 16-bit masks per word
        uint32_t cnt = 0;
        const uint32_t sp = tid * SPAN;
#pragma unroll
        for (uint32_t pc = 0; pc < PIECES; ++pc) {
            uint4 v = *(const uint4 *)(text + sp + pc * 16);
            uint32_t nl = eq_mask4(v.x, 0x0A0A0A0Au) | (eq_mask4(v.y, 0x0A0A0A0Au) << 4) | (eq_mask4(v.z, 0x0A0A0A0Au) << 8) | (eq_mask4(v.w, 0x0A0A0A0Au) << 12);
            uint32_t cr = eq_mask4(v.x, 0x0D0D0D0Du) | (eq_mask4(v.y, 0x0D0D0D0Du) << 4) | (eq_mask4(v.z, 0x0D0D0D0Du) << 8) | (eq_mask4(v.w, 0x0D0D0D0Du) << 12);
            while (cr) {                                                 // rare
                uint32_t b = __builtin_ctz(cr); cr &= cr - 1;
                uint32_t q = sp + pc * 16 + b;
                uint8_t nx = (q + 1 < TEXT) ? text[q + 1] : ((c0 + q + 1 < a.n_bytes) ? a.gaf[c0 + q + 1] : 0);
                if (nx != '\n') nl |= 1u << b;
            }
            // ignore anything at or beyond the valid length
            uint32_t base = sp + pc * 16;
            if (base >= V) nl = 0; else if (base + 16 > V) nl &= (1u << (V - base)) - 1u;
            if (pc & 1) mask[pc >> 1] |= nl << 16; else mask[pc >> 1] = nl;
            cnt += __builtin_popcount(nl);
        }
        // does the chunk begin at a line start?  (file start, or a terminator right before it)
        uint32_t head = 0;
        if (tid == 0) {
            if (c0 == 0) head = 1;
            else {
                uint8_t pb = a.gaf[c0 - 1];
                head = (pb == '\n') || (pb == '\r' && text[0] != '\n');
            }
        }
        uint32_t mine = cnt + head;
        uint32_t wtot, excl = wave_excl_scan(mine, wtot);
        if (lane == 63) misc[4 + wave] = wtot;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < WG / 64; ++w) { uint32_t x = misc[4 + w]; if (w < wave) wbase += x; total += x; }
        uint32_t o = wbase + excl;
        uint32_t owned = 0;
        const bool too_dense = total > MAXSTARTS;
        if (!too_dense) {
            if (head) { starts[o++] = 0; owned += (0 < V); }
#pragma unroll
            for (uint32_t pc = 0; pc < PIECES; ++pc) {
                uint32_t m = (pc & 1) ? (mask[pc >> 1] >> 16) : (mask[pc >> 1] & 0xFFFFu);
                while (m) {
                    uint32_t b = __builtin_ctz(m); m &= m - 1;
                    uint32_t st = sp + pc * 16 + b + 1;
                    starts[o++] = (uint16_t)st;
                    owned += (st < CHUNK && st < V);
                }
            }
        }
        if (owned) atomicAdd(&misc[0], owned);
        __syncthreads();
        const uint32_t n_owned = misc[0];
        if (too_dense) {
            // > CHUNK/24 lines in the stripe: some line is shorter than 12 columns -> the reference raises ValueError
            if (tid == 0) atomicMin(&a.st->err, ((a.base_offset + c0) << 3) | SVJG_EXC_VALUE_ERROR);
            __syncthreads();
            continue;
        }
        const bool at_eof = c0 + V == a.n_bytes;

        // ---- C + D: one line per lane, wave-level commit ----------------------------------------------
        for (uint32_t base = 0; base < n_owned; base += WG) {           // uniform trip count across the block
            uint32_t li = base + tid;
            int status = 1;                                              // 1 = no line
            uint32_t m = 0, s = 0;
            LaneList out{lists + tid};
            if (li < n_owned) {
                s = starts[li];
                uint32_t e;
                bool complete = true;
                if (li + 1 < total) e = (uint32_t)starts[li + 1] - 1;
                else if (at_eof) e = V;
                else { complete = false; e = V; }
                if (!complete || a.all_slow) status = -30;
                else status = fast_line(g, (const uint8_t *)text, s, e, out, HMAX, &m);
            }
            const bool have = li < n_owned;
            const bool defer = have && status < 0;
            // deferred lines: one atomic per wave
            unsigned long long db = __ballot(defer);
            if (db) {
                uint32_t nd = __popcll(db);
                unsigned long long dbase = 0;
                if (lane == 0) dbase = atomicAdd(&a.st->n_deferred, (unsigned long long)nd);
                dbase = __shfl(dbase, 0);
                if (defer) {
                    unsigned long long idx = dbase + __popcll(db & ((1ull << lane) - 1ull));
                    if (idx < a.deferred_cap) a.deferred[idx] = c0 + s; else atomicOr(&a.st->overflow, 1u);
                }
            }
            if (defer) m = 0;
            // counts: packed (ref | alt << 32) adds; neighbouring SVs of one alignment share cache lines
            for (uint32_t j = 0; j < m; ++j) {
                Pending h = out[j];
                atomicAdd(&a.counts[h.hit], (unsigned long long)(h.pre & 0xFFFFu) | ((unsigned long long)(h.pre >> 16) << 32));
            }
            if (a.want_hits) {
                uint32_t tot, ex = wave_excl_scan(m, tot);
                if (tot) {
                    unsigned long long rb = 0;
                    if (lane == 0) rb = atomicAdd(&a.st->n_recs, (unsigned long long)tot);
                    rb = __shfl(rb, 0);
                    for (uint32_t j = 0; j < m; ++j) {
                        unsigned long long idx = rb + ex + j;
                        if (idx < a.rec_cap) {
                            Pending h = out[j];
                            svjg_hitrec r; r.line_start = a.base_offset + c0 + s; r.slot = h.hit;
                            r.n_ref = (uint16_t)(h.pre & 0xFFFFu); r.n_alt = (uint16_t)(h.pre >> 16);
                            a.recs[idx] = r;
                        } else atomicOr(&a.st->overflow, 2u);
                    }
                }
            }
        }
        if (tid == 0) wg_lines += n_owned;
        __syncthreads();                                                 // LDS is reused by the next stripe
    }
    if (tid == 0 && wg_lines) atomicAdd(&a.st->n_lines, wg_lines);
}

struct SlowEmit {
    const ClassifyArgs *a;
    uint64_t line_start;
    __device__ void operator()(uint32_t slot, uint32_t allele) {
        atomicAdd(&a->counts[slot], allele ? (1ull << 32) : 1ull);
        if (a->want_hits) {
            unsigned long long idx = atomicAdd(&a->st->n_recs, 1ull);
            if (idx < a->rec_cap) {
                svjg_hitrec r; r.line_start = line_start; r.slot = slot; r.n_ref = allele ? 0 : 1; r.n_alt = allele ? 1 : 0;
                a->recs[idx] = r;
            } else atomicOr(&a->st->overflow, 2u);
        }
    }
};

__global__ __launch_bounds__(WG) void k_classify_slow(ClassifyArgs a, uint64_t n_def) {
    uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
    if (i >= n_def) return;
    uint64_t s = a.deferred[i], e = s;
    while (e < a.n_bytes && a.gaf[e] != '\n' && a.gaf[e] != '\r') ++e;
    SlowEmit em{&a, a.base_offset + s};
    int rc = slow_line(a.g, a.gaf, s, e, em);
    if (rc) atomicMin(&a.st->err, ((a.base_offset + s) << 3) | (unsigned long long)rc);
}

// ---------------------------------------------------------------------------------------------------
// genotype likelihoods
// ---------------------------------------------------------------------------------------------------

struct dd { double hi, lo; };

__device__ inline dd two_sum(double a, double b) {
    double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ inline dd dd_add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = two_sum(s.hi, s.lo);          // quick renormalisation (|lo| << |hi| here)
    s.lo += t.lo;
    return two_sum(s.hi, s.lo);
}
__device__ inline dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
__device__ inline int dd_cmp(dd a, dd b) { return a.hi < b.hi ? -1 : a.hi > b.hi ? 1 : a.lo < b.lo ? -1 : a.lo > b.lo ? 1 : 0; }

// table[i] = log10(i!) ; three small kernels: per-block scan, scan of block sums, add offsets
constexpr uint32_t LF_BLOCK = 1024;

__global__ __launch_bounds__(LF_BLOCK) void k_logfact_local(dd *tab, dd *bsum, uint32_t n) {
    __shared__ dd sh[LF_BLOCK];
    uint32_t i = blockIdx.x * LF_BLOCK + threadIdx.x;
    double v = (i >= 2 && i < n) ? log10((double)i) : 0.0;
    sh[threadIdx.x] = dd{v, 0.0};
    __syncthreads();
    for (uint32_t d = 1; d < LF_BLOCK; d <<= 1) {
        dd x = sh[threadIdx.x], y = dd{0.0, 0.0};
        if (threadIdx.x >= d) y = sh[threadIdx.x - d];
        __syncthreads();
        sh[threadIdx.x] = dd_add(x, y);
        __syncthreads();
    }
    if (i < n) tab[i] = sh[threadIdx.x];
    if (threadIdx.x == LF_BLOCK - 1) bsum[blockIdx.x] = sh[threadIdx.x];
}

__global__ void k_logfact_bsum(dd *bsum, uint32_t nb) {       // exclusive scan of block sums, one lane (nb is small)
    if (threadIdx.x || blockIdx.x) return;
    dd run{0.0, 0.0};
    for (uint32_t b = 0; b < nb; ++b) { dd t = bsum[b]; bsum[b] = run; run = dd_add(run, t); }
}

__global__ __launch_bounds__(LF_BLOCK) void k_logfact_add(dd *tab, const dd *bsum, uint32_t n) {
    uint32_t i = blockIdx.x * LF_BLOCK + threadIdx.x;
    if (i < n && blockIdx.x) tab[i] = dd_add(tab[i], bsum[blockIdx.x]);
}

struct GenoArgs {
    const unsigned long long *counts;
    const uint8_t *sv_type; const uint32_t *slot; const uint8_t *ok;
    uint64_t n_rows; uint32_t min_support;
    double l_ok, l_err, l_half;        // log10(1-e), log10(e), log10(1/2) computed by the host libm like CPython does
    const dd *logfact; uint32_t logfact_n;
    uint8_t *gt; int64_t *pl; uint32_t *raw; uint8_t *genotyped;
    unsigned int *max_n;               // k_geno_maxn output
};

// normalised counts (predict-genotype.py:327-338) and the rounded ones fed to comb()
__device__ inline void geno_counts(uint32_t type, uint32_t ref, uint32_t alt, double &c1, double &c2, uint32_t &r1, uint32_t &r2) {
    c1 = (double)ref; c2 = (double)alt;
    if (type == 0 && ref) c1 = (double)ref * 0.5;       // round(x/2, 1) is exact for halves
    if (type == 1 && alt) c2 = (double)alt * 0.5;
    r1 = (uint32_t)rint(c1); r2 = (uint32_t)rint(c2);   // int(round(c, 0)): half to even
}

__device__ inline bool geno_gate(const GenoArgs &a, uint64_t r, uint32_t &ref, uint32_t &alt) {
    ref = alt = 0;
    const uint32_t ok = a.ok[r];
    if (!(ok & 1u) || a.slot[r] == NONE32) return false;
    unsigned long long c = a.counts[a.slot[r]];
    ref = (uint32_t)c; alt = (uint32_t)(c >> 32);
    // sv_id is a key of the informative dict (:216): a key exists iff it has >= 1 informative alignment,
    // unless the caller says the slot itself proves presence (stand-alone run from a JSON, ok bit 1)
    return (ok & 2u) || (ref | alt) != 0;
}

__global__ __launch_bounds__(WG) void k_geno_maxn(GenoArgs a) {
    uint64_t r = (uint64_t)blockIdx.x * WG + threadIdx.x;
    uint32_t n = 0;
    if (r < a.n_rows) {
        uint32_t ref, alt;
        if (geno_gate(a, r, ref, alt)) { double c1, c2; uint32_t r1, r2; geno_counts(a.sv_type[r], ref, alt, c1, c2, r1, r2); n = r1 + r2; }
    }
    for (int d = 32; d; d >>= 1) { uint32_t y = __shfl_down(n, d); n = n > y ? n : y; }
    if ((threadIdx.x & 63) == 0 && n) atomicMax(a.max_n, n);
}

__device__ inline int64_t trunc_dd(dd v) {               // int(Decimal): toward zero
    double t = trunc(v.hi);
    if (t == v.hi) {                                     // hi is integral: the tail decides
        if (v.hi > 0 && v.lo < 0) t -= 1.0;
        else if (v.hi < 0 && v.lo > 0) t += 1.0;
    }
    return (int64_t)t;
}

__global__ __launch_bounds__(WG) void k_genotype(GenoArgs a) {
    uint64_t r = (uint64_t)blockIdx.x * WG + threadIdx.x;
    if (r >= a.n_rows) return;
    uint32_t ref, alt;
    bool go = geno_gate(a, r, ref, alt);
    a.raw[r * 2] = go ? ref : 0; a.raw[r * 2 + 1] = go ? alt : 0;
    a.genotyped[r] = go;
    if (!go) { a.gt[r] = 3; a.pl[r * 3] = a.pl[r * 3 + 1] = a.pl[r * 3 + 2] = 0; return; }
    double c1, c2; uint32_t r1, r2;
    geno_counts(a.sv_type[r], ref, alt, c1, c2, r1, r2);
    // products in double, sums exact (the reference adds Decimal images of the doubles, :295-297)
    dd l0 = two_sum(c1 * a.l_ok, c2 * a.l_err);
    dd l1 = dd{(c1 + c2) * a.l_half, 0.0};
    dd l2 = two_sum(c2 * a.l_ok, c1 * a.l_err);
    int c01 = dd_cmp(l0, l1), c02 = dd_cmp(l0, l2), c12 = dd_cmp(l1, l2);
    uint8_t g = 3;
    if (c01 > 0 && c02 > 0) g = 0; else if (c01 < 0 && c12 > 0) g = 1; else if (c02 < 0 && c12 < 0) g = 2;
    if (!(c1 + c2 >= (double)a.min_support)) g = 3;
    a.gt[r] = g;
    uint32_t n = r1 + r2;
    dd comb{0.0, 0.0};
    if (n < a.logfact_n) comb = dd_add(dd_add(a.logfact[n], dd_neg(a.logfact[n - r1])), dd_neg(a.logfact[r1]));
    comb = dd{comb.hi, 0.0};                             // the reference rounds log10(comb) to a double first (:313)
    dd ls[3] = {l0, l1, l2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dd s = dd_add(ls[i], comb);
        dd p = dd_add(dd_add(dd_add(s, s), dd_add(s, s)), s);             // 5 s
        p = dd_add(p, p);                                                 // 10 s
        a.pl[r * 3 + i] = trunc_dd(dd_neg(p));
    }
}

}  // namespace svjg
