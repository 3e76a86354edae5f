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
//   k_classify_slow  one lane per deferred line, exact string path (svjg::slow_line) on an LDS copy of the line
//   k_logfact_*      log10(i!) table in double-double for the binomial term
//   k_genotype       one VCF row per lane, fp64 / double-double likelihoods (predict-genotype.py:281-325)
#pragma once
#include <hip/hip_runtime.h>
#include "svjg_line.h"

namespace svjg {

#ifndef SVJG_WG
#define SVJG_WG 512
#define SVJG_PIECES 5
#endif
#ifndef SVJG_LRW
#define SVJG_LRW 32
#endif
#ifndef SVJG_NMAXW
#define SVJG_NMAXW 160
#endif
constexpr uint32_t WG = SVJG_WG;                 // classify kernel: 8 waves per workgroup, two workgroups per CU (LDS-bound) = 16 waves / CU
constexpr uint32_t NWAVE = WG / 64;
constexpr uint32_t TPB = 256;                    // block size of the small per-row / per-line kernels
constexpr uint32_t PIECES = SVJG_PIECES;         // 16-byte pieces of text per lane and stripe
constexpr uint32_t SPAN = PIECES * 16;           // bytes of byte classification per lane
constexpr uint32_t SLICE = SPAN * 64;            // bytes of text whose lines one wave owns
constexpr uint32_t TEXT = SPAN * WG;             // 40 KB staged in LDS
// A stripe = the bytes of text whose lines one workgroup iteration owns = TEXT minus a look-ahead that lets lines
// starting near its end be complete.  The look-ahead is a launch parameter (ClassifyArgs::chunk = TEXT - look-ahead): the
// host starts at LOOK_MIN and doubles it, up to LOOK_MAX, when a batch had many lines cut off by the staged text (they
// go to the exact path, which is correct but slow), so short-line files do not pay for long-line ones.
constexpr uint32_t LOOK_MIN = 2048, LOOK_MAX = 16384;
constexpr uint32_t MAXSTARTS = TEXT / 24 + 8;    // a valid line has >= 24 bytes incl. its terminator
constexpr uint32_t KMAX = 128;                   // path nodes per alignment handled by the main kernel (longer paths: exact path)
constexpr uint32_t LRW = SVJG_LRW;               // lines per wave and round (line-granular phases use the first LRW lanes)
constexpr uint32_t NMAXW = SVJG_NMAXW;           // path nodes per wave and round
static_assert(LRW <= 64 && KMAX <= NMAXW, "round geometry");
static_assert(TEXT < 65536, "text offsets are kept in 16 bits");

// LDS carve-up of k_classify_main (bytes): text and the two byte-class bitmaps are shared by the workgroup,
// everything else is private to one wave
constexpr uint32_t L_TEXT = 0;
constexpr uint32_t L_TABBM = L_TEXT + TEXT + 16;                           // u16[TEXT/16] one bit per byte: '\t'
constexpr uint32_t L_ORIBM = L_TABBM + TEXT / 8;                           // u16[TEXT/16] one bit per byte: '<' or '>'
constexpr uint32_t L_MISC = L_ORIBM + TEXT / 8 + 16;                       // u32[32]: [0] line starts in the stripe, [8 + w] first start found by wave w
constexpr uint32_t L_WAVE = L_MISC + 128;
constexpr uint32_t W_RSTART = 0;                                           // u16[LRW + 2]  line starts of the round (+ the end of the last line)
constexpr uint32_t W_TS = (W_RSTART + (LRW + 2) * 2 + 15) / 16 * 16;       // u32[LRW]  path start column
constexpr uint32_t W_TE = W_TS + LRW * 4;                                  // u32[LRW]
constexpr uint32_t W_TLEN = W_TE + LRW * 4;                                // u32[LRW]
constexpr uint32_t W_TOT = W_TLEN + LRW * 4;                               // u32[LRW]  sum of node lengths
constexpr uint32_t W_META = W_TOT + LRW * 4;                               // u32[LRW]  nbase | k << 16 | status << 24
constexpr uint32_t W_PBEG = W_META + LRW * 4;                              // u16[LRW]  tab before the path column
constexpr uint32_t W_PEND = W_PBEG + LRW * 2;                              // u16[LRW]  tab after the path column
constexpr uint32_t W_NPOS = W_PEND + LRW * 2;                              // u16[NMAXW]  name start (orientation mark + 1)
constexpr uint32_t W_NLINE = W_NPOS + NMAXW * 2;                           // u16[NMAXW]  line in round | orientation << 15
constexpr uint32_t W_NFIRST = W_NLINE + NMAXW * 2;                         // u16[NMAXW]  first node of the line with the same name
constexpr uint32_t W_NID = (W_NFIRST + NMAXW * 2 + 3) / 4 * 4;             // u32[NMAXW]
constexpr uint32_t W_NPRE = W_NID + NMAXW * 4;                             // u32[NMAXW]  length, then inclusive prefix
constexpr uint32_t WAVE_BYTES = (W_NPRE + NMAXW * 4 + 15) / 16 * 16;
constexpr uint32_t LDS_MAIN = L_WAVE + NWAVE * WAVE_BYTES;

// status words (device)
struct DevStatus {
    unsigned long long n_lines;
    unsigned long long n_deferred;       // entries appended to the deferred list
    unsigned long long n_recs;           // hit records appended
    unsigned long long err;              // min over (file offset << 3 | exception class); ~0 = none
    unsigned long long n_incomplete;     // lines deferred because they run past the staged text
    unsigned int non_ascii;
    unsigned int overflow;               // bit 0: deferred list, bit 1: hit-record buffer
};

struct ClassifyArgs {
    const uint8_t *gaf;                  // resident text, allocation padded with >= TEXT + 64 zero bytes
    uint64_t n_bytes;
    uint64_t base_offset;
    GraphView g;                         // global-memory views
    uint32_t all_slow;
    uint32_t want_hits;
    uint32_t n_chunks;
    uint32_t chunk;                      // stripe stride in bytes (multiple of 16, TEXT - look-ahead)
    uint32_t diag;                       // measurement only (SVJG_DIAG): 1 stop after B, 2 stop after R2, 4 no node lookup, 8 no atomics
    unsigned long long *counts;          // [n_slots] ref | alt << 32
    uint64_t *deferred;  uint64_t deferred_cap;
    svjg_hitrec *recs;   uint64_t rec_cap;
    DevStatus *st;
    unsigned long long *dbg;             // measurement only (SVJG_DIAG & 16): per-phase cycle sums of lane 0 of every workgroup
};

__device__ inline uint32_t zero_bytes(uint32_t t) {                    // 0x80 in every byte of t that is zero (exact)
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}
__device__ inline uint32_t movemask4(uint32_t m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }
__device__ inline uint32_t eq_mask16(uint4 v, uint32_t pat) {          // 16-bit mask of bytes equal to pat's byte
    // byte flags (0x80 / 0x00) gathered with two chained v_dot4_u32_u8 per half: weights 1,2,4,8 / 16,32,64,128
    const uint32_t lo = __builtin_amdgcn_udot4(zero_bytes(v.x ^ pat), 0x08040201u, __builtin_amdgcn_udot4(zero_bytes(v.y ^ pat), 0x80402010u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(zero_bytes(v.z ^ pat), 0x08040201u, __builtin_amdgcn_udot4(zero_bytes(v.w ^ pat), 0x80402010u, 0u, false), false);
    return (lo | (hi << 8)) >> 7;
}

// The same for text known to be pure ASCII (every byte < 0x80: no carries between the byte lanes of the SWAR add), four
// operations per word instead of six.  `pat` bytes must be < 0x80 too.
__device__ inline uint32_t ne_flags_ascii(uint32_t w, uint32_t pat) { return ((w ^ pat) + 0x7F7F7F7Fu) & 0x80808080u; }   // 0x80 where the byte differs
__device__ inline uint32_t eq_mask16_ascii(uint4 v, uint32_t pat) {
    const uint32_t lo = __builtin_amdgcn_udot4(ne_flags_ascii(v.x, pat), 0x08040201u, __builtin_amdgcn_udot4(ne_flags_ascii(v.y, pat), 0x80402010u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(ne_flags_ascii(v.z, pat), 0x08040201u, __builtin_amdgcn_udot4(ne_flags_ascii(v.w, pat), 0x80402010u, 0u, false), false);
    return ~((lo | (hi << 8)) >> 7) & 0xFFFFu;
}
template <bool ASCII> __device__ inline uint32_t eq_mask16_t(uint4 v, uint32_t pat) { return ASCII ? eq_mask16_ascii(v, pat) : eq_mask16(v, pat); }
template <bool ASCII> __device__ inline bool any_byte_t(uint4 v, uint32_t pat) {
    if (ASCII) {
        const uint32_t all = ((v.x ^ pat) + 0x7F7F7F7Fu) & ((v.y ^ pat) + 0x7F7F7F7Fu) & ((v.z ^ pat) + 0x7F7F7F7Fu) & ((v.w ^ pat) + 0x7F7F7F7Fu);
        return (~all & 0x80808080u) != 0;
    }
    return (zero_bytes(v.x ^ pat) | zero_bytes(v.y ^ pat) | zero_bytes(v.z ^ pat) | zero_bytes(v.w ^ pat)) != 0;
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

// block-wide exclusive scan; `slot` = WG/64 words of LDS; two barriers inside
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *slot, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t wtot, ex = wave_excl_scan(v, wtot);
    __syncthreads();
    if (lane == 63) slot[wave] = wtot;
    __syncthreads();
    uint32_t base = 0; total = 0;
#pragma unroll
    for (uint32_t w = 0; w < WG / 64; ++w) { uint32_t x = slot[w]; if (w < wave) base += x; total += x; }
    return base + ex;
}

enum : uint32_t { ST_NONE = 0, ST_OK = 1, ST_NOHIT = 2, ST_DEFER = 3 };   // per-line status inside a round

// plain decimal column text[a, b): 1..9 digits and nothing else -> value.  Straight-line SWAR: three aligned
// LDS words, digits checked and folded pairwise (no per-digit loop, no divergence).
__device__ inline bool field_dec(const uint8_t *text, uint32_t a, uint32_t b, uint32_t &v) {
    const uint32_t n = b - a;
    const uint32_t *w = (const uint32_t *)(text + (a & ~3u));
    const uint32_t sh = a & 3u, d0 = w[0], d1 = w[1], d2 = w[2];
    uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh) ^ 0x30303030u;    // chars a .. a+3 as digits
    uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sh) ^ 0x30303030u;    // chars a+4 .. a+7
    const uint32_t n8 = n < 8 ? n : 8;
    // keep the first n8 bytes, then shift them to the top of the 64-bit (hi:lo) so that leading bytes are zero digits
    const uint32_t klo = n8 >= 4 ? 0xFFFFFFFFu : ((1u << ((8 * n8) & 31u)) - 1u);
    const uint32_t khi = n8 >= 8 ? 0xFFFFFFFFu : (n8 > 4 ? ((1u << ((8 * (n8 - 4)) & 31u)) - 1u) : 0u);
    lo &= klo; hi &= khi;
    bool ok = (((lo + 0x76767676u) | lo | (hi + 0x76767676u) | hi) & 0x80808080u) == 0;     // every kept byte is 0..9
    const unsigned long long x = (((unsigned long long)hi << 32) | lo) << ((8 * (8 - n8)) & 63u);
    lo = (uint32_t)x; hi = (uint32_t)(x >> 32);
    uint32_t pl = (lo * 10u + (lo >> 8)) & 0x00FF00FFu, ph = (hi * 10u + (hi >> 8)) & 0x00FF00FFu;
    uint32_t r = ((pl & 0xFFu) * 100u + (pl >> 16)) * 10000u + (ph & 0xFFu) * 100u + (ph >> 16);
    if (n == 9) { uint32_t d = (uint32_t)text[a + 8] - '0'; ok &= d <= 9; r = r * 10u + d; }
    v = r;
    return ok & (n - 1u <= 8u);
}

// next set bit of a bitmap, scanning upwards from the cursor (word index wi, remaining bits `cur`); `lim` = end position
struct BitCursor {
    const uint32_t *bm; uint32_t wi, cur, lim;
    __device__ uint32_t next() {                                         // position of the next set bit, or lim
        while (cur == 0) { ++wi; if ((wi << 5) >= lim) return lim; cur = bm[wi]; }
        uint32_t b = __builtin_ctz(cur); cur &= cur - 1;
        uint32_t p = (wi << 5) + b;
        return p < lim ? p : lim;
    }
};

// Path segment text[a0, a0+L), 1 <= L <= 32: its eight zero-padded words -> d, and the pre-hash of the node-name table
// (svjg_host_tables.h: name_prehash_host).
__device__ inline uint32_t name_words(const uint8_t *text, uint32_t a0, uint32_t L, uint32_t d[8]) {
    const uint32_t *w = (const uint32_t *)(text + (a0 & ~3u));
    const uint32_t sh = a0 & 3u;
    uint32_t prev = w[0];
    uint32_t h = L * 0x7FEB352Du;
    const uint32_t C[8] = {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u};
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t nx = w[i + 1];
        const uint32_t nb = L > 4 * i ? L - 4 * i : 0u;                 // bytes of the name in this word
        d[i] = __builtin_amdgcn_alignbyte(nx, prev, sh) & (nb >= 4 ? 0xFFFFFFFFu : ((1u << ((8 * nb) & 31u)) - 1u));
        prev = nx;
        h += d[i] * C[i];
    }
    return h;
}

// the two candidate slots of a pre-hash in a two-choice table (svjg_host_tables.h: cuckoo_slots_host)
__device__ inline void cuckoo_slots(uint32_t x, uint32_t seed, uint32_t mask, uint32_t &s1, uint32_t &s2) {
    uint32_t p = x ^ seed;
    p ^= p >> 15; p *= 0x2C1B3C6Du; p ^= p >> 12;
    uint32_t q = (x + seed) * 0x85EBCA6Bu;
    q ^= q >> 13; q *= 0xC2B2AE35u; q ^= q >> 16;
    s1 = p & mask; s2 = q & mask;
    if (s2 == s1) s2 = s1 ^ 1u;
}

// entry of the node-name table (svjg_host_tables.h): q0 = name bytes 0..15, q1 = bytes 16..23 | meta | length in bp,
// q2 (fetched for names longer than 24 bytes only) = bytes 24..31
__device__ inline bool name_match(const uint4 q0, const uint4 q1, const uint4 q2, const uint32_t d[8], uint32_t L) {
    return (q1.z & 31u) == L - 1u && q0.x == d[0] && q0.y == d[1] && q0.z == d[2] && q0.w == d[3] &&
           q1.x == d[4] && q1.y == d[5] && (L <= 24u || (q2.x == d[6] && q2.y == d[7]));
}

__device__ inline uint32_t link_prehash(uint32_t klo, uint32_t khi) { return klo ^ (khi * 0x9E3779B1u); }

#ifndef SVJG_UB3
#define SVJG_UB3 1
#endif
#ifndef SVJG_UB5
#define SVJG_UB5 2
#endif
constexpr uint32_t UB3 = SVJG_UB3, UB5 = SVJG_UB5;  // path nodes (R3) / path steps (R5) per lane handled at a time (table loads in flight)

// LDS traffic between lanes of ONE wave: DS operations of a wave execute in order, the fences only pin the compiler
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The classify kernel.  One workgroup walks stripes of the GAF text; the next stripe's HBM loads are issued into
// registers before the current one is processed, so the only HBM read of the text overlaps the parse.  Per stripe:
//   A  registers -> LDS (16 B per lane, coalesced on the HBM side)                                   [workgroup barrier]
//   B  one SPAN per lane, branch-free SWAR classification of every byte: tab and orientation-mark ('<' '>') bitmaps
//      -> LDS, line terminators -> per-lane masks, wave prefix sum                                    [workgroup barrier]
//   then every WAVE on its own (no workgroup barriers: the sixteen waves of a CU drift through different phases, so
//   table latency of one overlaps the parsing of another), for the lines whose preceding terminator lies in the wave's
//   SLICE of the stripe, in rounds of up to LRW lines / NMAXW path nodes:
//   R1 one LINE per lane: the twelve column boundaries by bit-scanning the tab bitmap, the nine decimal columns by
//      SWAR, path geometry; wave prefix sum hands every line a contiguous range of node slots
//   R2 one LINE per lane: node slots filled from the orientation bitmap (name position, line, orientation)
//   R3 one path NODE per lane: name -> slot of the node-name hash table (Infinity Cache / L2 resident) -> id, length
//   R4 one LINE per lane: running path length, first occurrence of every name (the reference's list.index /
//      str.split quirks), validation
//   R5 one path STEP (link) per lane: overlap test on the prefix sums, link hash table lookup, one 64-bit
//      atomic (ref | alt << 32) per hit, optional hit records
//   R6 deferred-line offsets, one aggregated atomic per wave
//                                                                                                     [workgroup barrier]
// Phase B of k_classify_main for one lane: classes of the SPAN bytes at text + tid * SPAN.  Tab and orientation-mark
// bitmaps -> LDS; line terminators -> `mask` (two 16-bit masks per word), their number -> cnt, and how many of the
// lines they start belong to this stripe -> owned.  terminator = '\n', or a '\r' not followed by '\n' (Python
// universal newlines).
template <bool ASCII>
__device__ inline void classify_span(const ClassifyArgs &a, const uint8_t *text, uint16_t *tabbm16, uint16_t *oribm16, uint32_t tid,
                                     uint64_t c0, uint32_t V, uint32_t own_lim, uint32_t (&mask)[(PIECES + 1) / 2], uint32_t &cnt, uint32_t &owned) {
    const uint32_t sp = tid * SPAN;
#pragma unroll
    for (uint32_t pc = 0; pc < PIECES; ++pc) {
        const uint4 v = *(const uint4 *)(text + sp + pc * 16);
        uint32_t nl = eq_mask16_t<ASCII>(v, 0x0A0A0A0Au);
        tabbm16[tid * PIECES + pc] = (uint16_t)eq_mask16_t<ASCII>(v, 0x09090909u);
        oribm16[tid * PIECES + pc] = (uint16_t)eq_mask16_t<ASCII>(make_uint4(v.x | 0x02020202u, v.y | 0x02020202u, v.z | 0x02020202u, v.w | 0x02020202u), 0x3E3E3E3Eu);
        // carriage returns: cheap any-test first (no text file has them in practice)
        if (any_byte_t<ASCII>(v, 0x0D0D0D0Du)) {
            uint32_t cr = eq_mask16(v, 0x0D0D0D0Du);
            while (cr) {
                uint32_t b = __builtin_ctz(cr); cr &= cr - 1;
                uint32_t q = sp + pc * 16 + b;
                uint8_t nx = (q + 1 < TEXT) ? text[q + 1] : ((c0 + q + 1 < a.n_bytes) ? a.gaf[c0 + q + 1] : 0);
                if (nx != '\n') nl |= 1u << b;
            }
        }
        const uint32_t pb = sp + pc * 16;                            // ignore anything at or beyond the valid length
        if (pb >= V) nl = 0; else if (pb + 16 > V) nl &= (1u << (V - pb)) - 1u;
        if (pc & 1) mask[pc >> 1] |= nl << 16; else mask[pc >> 1] = nl;
        cnt += __builtin_popcount(nl);
        // a terminator at pb + b starts a line at pb + b + 1; the stripe owns it if that is below own_lim
        const uint32_t keep = own_lim > pb + 1 ? (own_lim - pb - 1 < 16u ? own_lim - pb - 1 : 16u) : 0u;
        owned += __builtin_popcount(nl & ((1u << keep) - 1u));
    }
}

#ifndef SVJG_MINW
#define SVJG_MINW ((2 * SVJG_WG) / 256)
#endif
__global__ __launch_bounds__(WG, SVJG_MINW) void k_classify_main(ClassifyArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *text = lds + L_TEXT;
    uint16_t *tabbm16 = (uint16_t *)(lds + L_TABBM), *oribm16 = (uint16_t *)(lds + L_ORIBM);
    const uint32_t *tabbm = (const uint32_t *)(lds + L_TABBM), *oribm = (const uint32_t *)(lds + L_ORIBM);
    uint32_t *misc = (uint32_t *)(lds + L_MISC);

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t *wb = lds + L_WAVE + wave * WAVE_BYTES;                    // this wave's private arrays
    uint16_t *rstart = (uint16_t *)(wb + W_RSTART);
    uint32_t *l_ts = (uint32_t *)(wb + W_TS), *l_te = (uint32_t *)(wb + W_TE), *l_tlen = (uint32_t *)(wb + W_TLEN);
    uint32_t *l_tot = (uint32_t *)(wb + W_TOT), *l_meta = (uint32_t *)(wb + W_META);
    uint16_t *l_pbeg = (uint16_t *)(wb + W_PBEG), *l_pend = (uint16_t *)(wb + W_PEND);
    uint16_t *n_pos = (uint16_t *)(wb + W_NPOS), *n_line = (uint16_t *)(wb + W_NLINE), *n_first = (uint16_t *)(wb + W_NFIRST);
    uint32_t *n_id = (uint32_t *)(wb + W_NID), *n_pre = (uint32_t *)(wb + W_NPRE);

    const GraphView g = a.g;

    unsigned long long wave_lines = 0;

    // stripe prefetch registers
    uint4 pf[PIECES];
    uint32_t pf_head = '\n';                                           // byte right before the stripe (decides whether it starts a line)
    auto prefetch = [&](uint32_t chunk) {                            // no bounds tests: the buffer is zero padded by TEXT + 64 bytes
        const uint64_t c0 = (uint64_t)(chunk < a.n_chunks ? chunk : a.n_chunks - 1) * a.chunk;
        const uint4 *src = (const uint4 *)(a.gaf + c0) + tid;
#pragma unroll
        for (uint32_t i = 0; i < PIECES; ++i) pf[i] = src[i * WG];
        pf_head = c0 ? a.gaf[c0 - 1] : (uint32_t)'\n';
    };
    prefetch(blockIdx.x);
    if (tid == 0) misc[1] = 0;                                           // "stripe holds a byte >= 0x80"
    __syncthreads();

    for (uint32_t chunk = blockIdx.x; chunk < a.n_chunks; chunk += gridDim.x) {
        const uint64_t c0 = (uint64_t)chunk * a.chunk;
        const uint32_t V = (uint32_t)((a.n_bytes - c0 < (uint64_t)TEXT) ? (a.n_bytes - c0) : (uint64_t)TEXT);   // valid bytes staged
        const uint32_t own_lim = V < a.chunk ? V : a.chunk;                  // lines starting below this offset belong to the stripe

        // ---- A: registers -> LDS, then start the next stripe's HBM loads ---------------------------------
        uint32_t hi_bits = 0;
#pragma unroll
        for (uint32_t i = 0; i < PIECES; ++i) {
            hi_bits |= pf[i].x | pf[i].y | pf[i].z | pf[i].w;
            *(uint4 *)(text + (i * WG + tid) * 16) = pf[i];
        }
        if (tid == 0) misc[0] = 0;
        if (hi_bits & 0x80808080u) { a.st->non_ascii = 1; misc[1] = 1; }
        const uint32_t head_byte = pf_head;
        __syncthreads();
        const bool ascii = misc[1] == 0;                                 // workgroup-uniform: the cheaper SWAR classes apply
        prefetch(chunk + gridDim.x);

        // ---- B: byte classes ----------------------------------------------------------------------------
        uint32_t mask[(PIECES + 1) / 2];                                 // two 16-bit terminator masks per word
        uint32_t cnt = 0, owned = 0;
        const uint32_t sp = tid * SPAN;
        if (ascii) classify_span<true>(a, text, tabbm16, oribm16, tid, c0, V, own_lim, mask, cnt, owned);
        else classify_span<false>(a, text, tabbm16, oribm16, tid, c0, V, own_lim, mask, cnt, owned);
        uint32_t head = 0;                                               // does the stripe begin at a line start?
        if (tid == 0) { head = (head_byte == '\n') || (head_byte == '\r' && text[0] != '\n'); owned += head & (0 < own_lim); }
        // lines belong to the wave that sees the terminator in front of them: one wave prefix sum, no workgroup scan
        uint32_t wsum;
        const uint32_t sc = wave_excl_scan((cnt + head) | (owned << 16), wsum);
        const uint32_t o = sc & 0xFFFFu;                                 // starts found by lower lanes of the wave
        const uint32_t wtot = __builtin_amdgcn_readfirstlane(wsum & 0xFFFFu), n_w = __builtin_amdgcn_readfirstlane(wsum >> 16);
        if (o == 0 && cnt + head) {                                      // the wave's first start (ends the previous wave's last line)
            uint32_t first = 0;
            if (!head) {
#pragma unroll
                for (uint32_t pc = PIECES; pc-- > 0;) {
                    const uint32_t m = (pc & 1) ? (mask[pc >> 1] >> 16) : (mask[pc >> 1] & 0xFFFFu);
                    if (m) first = sp + pc * 16 + __builtin_ctz(m) + 1;
                }
            }
            misc[8 + wave] = first;
        }
        if (lane == 0) { if (wtot) atomicAdd(&misc[0], wtot); else misc[8 + wave] = 0xFFFFu; }
        __syncthreads();
        if (tid == 0) misc[1] = 0;                                       // (every wave has read it; set again only after the stripe's last barrier)
        if (misc[0] > MAXSTARTS) {
            // more than TEXT/24 lines in the stripe: some line has fewer than 12 columns -> ValueError in the reference
            if (tid == 0) atomicMin(&a.st->err, ((a.base_offset + c0) << 3) | SVJG_EXC_VALUE_ERROR);
            __syncthreads();
            continue;
        }
        const bool at_eof = c0 + V == a.n_bytes;
        uint32_t nxt = 0xFFFFu;                                          // first line start found by a later wave (0xFFFF: none in the staged text)
        for (uint32_t w = NWAVE - 1; w > wave; --w) { const uint32_t f = misc[8 + w]; if (f != 0xFFFFu) nxt = f; }

        for (uint32_t base = (a.diag & 1u) ? n_w : 0u, taken = 0; base < n_w; base += taken) {   // wave-uniform trip count
            // ---- line starts of the round: rstart[i] = start of line base + i, rstart[count] = where the last one ends ----------
            {
                uint32_t idx = o;
                if (head) { if (idx - base <= LRW) rstart[idx - base] = 0; ++idx; }
#pragma unroll
                for (uint32_t pc = 0; pc < PIECES; ++pc) {
                    uint32_t m = (pc & 1) ? (mask[pc >> 1] >> 16) : (mask[pc >> 1] & 0xFFFFu);
                    while (m) {
                        const uint32_t b = __builtin_ctz(m); m &= m - 1;
                        if (idx - base <= LRW) rstart[idx - base] = (uint16_t)(sp + pc * 16 + b + 1);
                        ++idx;
                    }
                }
                if (lane == 0 && wtot - base <= LRW) rstart[wtot - base] = (uint16_t)nxt;
            }
            wave_sync();
            // ---- R1: one line per lane --------------------------------------------------------------
            const uint32_t li = base + lane;
            uint32_t status = ST_NONE, k = 0, s = 0;
            bool cut = false;                                            // the line runs past the staged text
            if (lane < LRW && li < n_w) {
                s = rstart[lane];
                const uint32_t nx = rstart[lane + 1];
                uint32_t e = V;
                bool complete = true;
                if (nx != 0xFFFFu) e = nx - 1;
                else if (!at_eof) { complete = false; cut = true; }
                status = ST_DEFER;
                if (complete && !a.all_slow) {
                    while (e > s && py_space(text[e - 1])) --e;         // line.rstrip()
                    BitCursor tc{tabbm, s >> 5, 0, e};
                    tc.cur = tabbm[s >> 5] & (0xFFFFFFFFu << (s & 31));
                    const uint32_t t0 = tc.next(), t1 = tc.next(), t2 = tc.next(), t3 = tc.next(), t4 = tc.next(), t5 = tc.next();
                    const uint32_t t6 = tc.next(), t7 = tc.next(), t8 = tc.next(), t9 = tc.next(), t10 = tc.next(), t11 = tc.next();
                    bool ok = t10 < e;                                   // twelve columns
                    uint32_t qlen, qs, qe, tlen, ts, te, am, alen, aq;
                    ok &= field_dec(text, t0 + 1, t1, qlen) & field_dec(text, t1 + 1, t2, qs) & field_dec(text, t2 + 1, t3, qe);
                    ok &= field_dec(text, t5 + 1, t6, tlen) & field_dec(text, t6 + 1, t7, ts) & field_dec(text, t7 + 1, t8, te);
                    ok &= field_dec(text, t8 + 1, t9, am) & field_dec(text, t9 + 1, t10, alen) & field_dec(text, t10 + 1, t11, aq);
                    ok &= alen != 0;                                     // ZeroDivisionError unless an id:f: tag exists: exact path decides
                    // path column (t4, t5): starts with an orientation mark; count the marks
                    const uint32_t pa = t4 + 1, pbnd = t5;
                    if (ok && pa < pbnd) {
                        const uint32_t w0 = pa >> 5, w1 = (pbnd - 1) >> 5;
                        for (uint32_t w = w0; w <= w1; ++w) {
                            uint32_t m = oribm[w];
                            if (w == w0) m &= 0xFFFFFFFFu << (pa & 31);
                            if (w == w1 && ((pbnd & 31) != 0)) m &= (1u << (pbnd & 31)) - 1u;
                            k += __builtin_popcount(m);
                        }
                        ok &= ((oribm[w0] >> (pa & 31)) & 1u) != 0;
                    } else ok = false;
                    ok &= k >= 1 && k <= KMAX;
                    if (ok) {
                        status = k >= 2 ? ST_OK : ST_NOHIT;
                        l_ts[lane] = ts; l_te[lane] = te; l_tlen[lane] = tlen; l_pbeg[lane] = (uint16_t)t4; l_pend[lane] = (uint16_t)t5;
                    }
                }
            }
            if (status != ST_OK) k = 0;
            uint32_t ntot;
            const uint32_t nbase = wave_excl_scan(k, ntot);
            // node slots exhausted: the round ends in front of the first line that does not fit (it opens the next round)
            const unsigned long long over = __ballot(nbase + k > NMAXW);
            const uint32_t avail = n_w - base < LRW ? n_w - base : LRW;
            taken = over ? (uint32_t)__builtin_ctzll(over) : avail;
            if (taken > avail) taken = avail;
            if (lane >= taken) { status = ST_NONE; k = 0; }
            const uint32_t n_nodes = (uint32_t)__shfl(nbase + k, (int)taken - 1);
            // ---- R2: node slots from the orientation bitmap; every name must be non-empty ------------------------
            if (k) {
                const uint32_t pa = (uint32_t)l_pbeg[lane] + 1, pbnd = l_pend[lane];
                BitCursor oc{oribm, pa >> 5, 0, pbnd};
                oc.cur = oribm[pa >> 5] & (0xFFFFFFFFu << (pa & 31));
                uint32_t prev = pa - 1;
                bool ok = true;
                for (uint32_t j = 0; j < k; ++j) {
                    const uint32_t q = oc.next();
                    ok &= (j == 0) | (q > prev + 1);
                    prev = q;
                    n_pos[nbase + j] = (uint16_t)(q + 1);
                    n_line[nbase + j] = (uint16_t)lane;                 // orientation bit added by the node's lane (R3)
                }
                ok &= pbnd > prev + 1;
                if (!ok) status = ST_DEFER;
            }
            if (lane < LRW) l_meta[lane] = nbase | (k << 16) | (status << 24);
            wave_sync();
            if (a.diag & 2u) continue;                                   // measurement only: stop after R2
            // ---- R3: one node per lane: hash the name, fetch BOTH candidate entries of the two-choice name table at once
            //      (one round trip for every lane, no probe sequences), compare the spelling --------------------------------
            for (uint32_t nb = 0; nb < n_nodes; nb += UB3 * 64) {
                uint32_t nn[UB3], lnv[UB3], d[UB3][8], len[UB3];
                uint4 e0[UB3][2], e1[UB3][2], e2[UB3][2];
                bool live[UB3], probe[UB3];
#pragma unroll
                for (uint32_t u = 0; u < UB3; ++u) {
                    nn[u] = nb + u * 64 + lane;
                    live[u] = false; probe[u] = false; lnv[u] = 0; len[u] = 0;
                    uint32_t s1 = 0, s2 = 0;
                    if (nn[u] < n_nodes) {
                        lnv[u] = n_line[nn[u]] & 0x7FFFu;
                        const uint32_t meta = l_meta[lnv[u]];
                        if ((meta >> 24) == ST_OK) {
                            live[u] = true;
                            const uint32_t a0 = n_pos[nn[u]];
                            const uint32_t lnb = meta & 0xFFFFu, lk = (meta >> 16) & 0xFFu;
                            const uint32_t b0 = (nn[u] + 1 < lnb + lk) ? (uint32_t)n_pos[nn[u] + 1] - 1u : (uint32_t)l_pend[lnv[u]];
                            if (text[a0 - 1] == '<') n_line[nn[u]] = (uint16_t)(lnv[u] | 0x8000u);
                            len[u] = b0 - a0;
                            probe[u] = len[u] - 1u <= 31u && !(a.diag & 4u);     // names of 1..32 bytes; longer ones: exact path
                            if (probe[u]) cuckoo_slots(name_words(text, a0, len[u], d[u]), g.name_seed, g.name_mask, s1, s2);
                        }
                    }
#pragma unroll
                    for (uint32_t c = 0; c < 2; ++c) {
                        e0[u][c] = e2[u][c] = make_uint4(0, 0, 0, 0); e1[u][c] = make_uint4(0, 0, 0xFFFFFFFFu, 0);
                        if (probe[u]) {
                            const uint4 *e = (const uint4 *)(g.name_tab + (size_t)(c ? s2 : s1) * 16);
                            e0[u][c] = e[0]; e1[u][c] = e[1];
                            if (len[u] > 24u) e2[u][c] = e[2];
                        }
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < UB3; ++u) {
                    if (live[u]) {
                        uint32_t id = NONE32, lbp = 0;
                        if (probe[u]) {
                            const bool m0 = name_match(e0[u][0], e1[u][0], e2[u][0], d[u], len[u]);
                            const bool m1 = name_match(e0[u][1], e1[u][1], e2[u][1], d[u], len[u]);
                            const uint32_t mz = m0 ? e1[u][0].z : e1[u][1].z, mw = m0 ? e1[u][0].w : e1[u][1].w;
                            // id << 7 | flags << 5 | byte length - 1, length in bp; hazard-prone name / unknown alt length: exact path
                            if ((m0 | m1) && mz != 0xFFFFFFFFu && !(mz & 0x60u)) { id = mz >> 7; lbp = mw; }
                        }
                        if (a.diag & 4u) { id = 0; lbp = 100; }
                        n_id[nn[u]] = id; n_pre[nn[u]] = lbp;
                        if (id == NONE32) atomicOr(&l_meta[lnv[u]], ST_DEFER << 24);              // ST_OK | ST_DEFER == ST_DEFER
                    }
                }
            }
            wave_sync();
            if (a.diag & 64u) continue;                                  // measurement only: stop after R3
            // ---- R4: one line per lane: prefix sums, first occurrences --------------------------------------
            if (lane < taken) {
                const uint32_t meta = l_meta[lane];
                if ((meta >> 24) == ST_OK) {
                    const uint32_t lnb = meta & 0xFFFFu, lk = (meta >> 16) & 0xFFu;
                    unsigned long long run = 0, seen1 = 0, seen2 = 0;
                    for (uint32_t j = 0; j < lk; ++j) {
                        const uint32_t idj = n_id[lnb + j];
                        run += n_pre[lnb + j];
                        n_pre[lnb + j] = (uint32_t)run;
                        uint32_t f = j;
                        // two-hash filter: only a possible revisit pays for the search of the first occurrence
                        const unsigned long long b1 = 1ull << (idj & 63), b2 = 1ull << ((idj * 0x9E3779B1u) >> 26);
                        if ((seen1 & b1) && (seen2 & b2))
                            for (uint32_t jj = 0; jj < j; ++jj) if (n_id[lnb + jj] == idj) { f = jj; break; }
                        seen1 |= b1; seen2 |= b2;
                        n_first[lnb + j] = (uint16_t)(lnb + f);
                    }
                    l_tot[lane] = (uint32_t)run;
                    if (run > 0xFFFFFFFFull) l_meta[lane] = (meta & 0x00FFFFFFu) | (ST_DEFER << 24);
                }
            }
            wave_sync();
            // ---- R5: one path step per lane; UB5 steps per lane, both candidate link-table entries of every step fetched in one round trip ------------
            for (uint32_t nb = 0; nb < n_nodes; nb += UB5 * 64) {
                uint32_t lnv[UB5], klo[UB5], khi[UB5], sa[UB5], sb[UB5];
                uint4 ek[UB5], ek2[UB5];
                bool go[UB5];
#pragma unroll
                for (uint32_t u = 0; u < UB5; ++u) {
                    const uint32_t n = nb + u * 64 + lane;
                    go[u] = false; lnv[u] = 0; klo[u] = khi[u] = sa[u] = sb[u] = 0;
                    if (n + 1 < n_nodes) {
                        const uint32_t ln = n_line[n] & 0x7FFFu;
                        const uint32_t meta = l_meta[ln];
                        const uint32_t lnb = meta & 0xFFFFu, lk = (meta >> 16) & 0xFFu;
                        if ((meta >> 24) == ST_OK && n + 1 < lnb + lk) {
                            // the reference evaluates the link (name, strand) of the FIRST occurrence of each name
                            // (str.split / list.index, filter-alignments.py:206, :269-271)
                            const uint32_t fl = n_first[n], fr = n_first[n + 1];
                            const long long left = (long long)n_pre[fl] - (long long)l_ts[ln];
                            const long long pre_excl_r = fr > lnb ? (long long)n_pre[fr - 1] : 0;
                            const long long right = (long long)l_tot[ln] - pre_excl_r - ((long long)l_tlen[ln] - (long long)l_te[ln] - 1);
                            if (left >= (long long)g.d_over && right >= (long long)g.d_over) {
                                go[u] = true; lnv[u] = ln;
                                klo[u] = (n_id[n + 1] << 1) | (uint32_t)(n_line[fr] >> 15);
                                khi[u] = (n_id[n] << 1) | (uint32_t)(n_line[fl] >> 15);
                                cuckoo_slots(link_prehash(klo[u], khi[u]), g.link_seed, g.link_mask, sa[u], sb[u]);
                            }
                        }
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < UB5; ++u) {
                    ek[u] = ek2[u] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0);
                    if (go[u]) { ek[u] = *(const uint4 *)(g.link_tab + (size_t)sa[u] * 4); ek2[u] = *(const uint4 *)(g.link_tab + (size_t)sb[u] * 4); }
                }
#pragma unroll
                for (uint32_t u = 0; u < UB5; ++u) {
                    uint32_t nh = 0, ea = 0, eb = 0;
                    bool many = false;
                    if (go[u]) {
                        if (!(ek[u].x == klo[u] && ek[u].y == khi[u])) ek[u] = ek2[u];                 // the other candidate slot
                        if (ek[u].x == klo[u] && ek[u].y == khi[u]) {
                            // one hit: (hit, NO_HIT); two: (hit, hit); more: (MANY | index into hits[], count)
                            ea = ek[u].z; eb = ek[u].w;
                            many = (ea & 0x80000000u) && eb != 0xFFFFFFFFu && ea != 0xFFFFFFFFu;
                            nh = many ? eb : (eb == 0xFFFFFFFFu ? 1u : 2u);
                        }
                    }
                    // hit records: one aggregated atomic per wave reserves the slots
                    unsigned long long rbase = 0;
                    if (a.want_hits) {
                        uint32_t wtot2, ex = wave_excl_scan(nh, wtot2);
                        if (wtot2) {
                            if (lane == 0) rbase = atomicAdd(&a.st->n_recs, (unsigned long long)wtot2);
                            rbase = __shfl(rbase, 0) + ex;
                        }
                    }
                    for (uint32_t j = 0; j < nh; ++j) {
                        const uint32_t hv = many ? g.hits[(ea & 0x7FFFFFFFu) + j] : (j == 0 ? ea : eb);
                        if (!(a.diag & 8u)) atomicAdd(&a.counts[hv >> 1], (hv & 1u) ? (1ull << 32) : 1ull);
                        if (a.want_hits) {
                            if (rbase + j < a.rec_cap) {
                                svjg_hitrec r; r.line_start = a.base_offset + c0 + rstart[lnv[u]]; r.slot = hv >> 1;
                                r.n_ref = (hv & 1u) ? 0 : 1; r.n_alt = (hv & 1u) ? 1 : 0;
                                a.recs[rbase + j] = r;
                            } else atomicOr(&a.st->overflow, 2u);
                        }
                    }
                }
            }
            // ---- R6: lines for the exact path ------------------------------------------------------------------
            {
                const bool defer = lane < taken && (l_meta[lane] >> 24) == ST_DEFER;
                const unsigned long long cb = __ballot(cut && lane < taken);
                if (cb && lane == 0) atomicAdd(&a.st->n_incomplete, (unsigned long long)__popcll(cb));
                unsigned long long db = __ballot(defer);
                if (db) {
                    unsigned long long dbase = 0;
                    if (lane == 0) dbase = atomicAdd(&a.st->n_deferred, (unsigned long long)__popcll(db));
                    dbase = __shfl(dbase, 0);
                    if (defer) {
                        unsigned long long idx = dbase + __popcll(db & ((1ull << lane) - 1ull));
                        if (idx < a.deferred_cap) a.deferred[idx] = c0 + s; else atomicOr(&a.st->overflow, 1u);
                    }
                }
            }
            wave_sync();                                                 // round state is reused
        }
        wave_lines += n_w;
        __syncthreads();                                                 // text and bitmaps are overwritten by the next stripe
    }
    if (lane == 0 && wave_lines) atomicAdd(&a.st->n_lines, wave_lines);
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

// The exact path: one lane per deferred line (svjg::slow_line, the reference's string semantics).  slow_line indexes
// the text byte by byte, so every wave first packs its 64 lines into LDS (each lane finds its line's terminator and
// copies the line, 16 bytes per step) and the string logic then pays LDS latency per byte, not HBM latency.  A line
// that does not fit (longer than SLOW_MAXLINE, or the 64 lines together exceed the buffer) is read in place.
constexpr uint32_t SLOW_TPB = 64, SLOW_LDS = 32 * 1024, SLOW_MAXLINE = 16 * 1024;
__global__ __launch_bounds__(SLOW_TPB) void k_classify_slow(ClassifyArgs a, uint64_t n_def) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[SLOW_LDS];
    const uint32_t lane = threadIdx.x;
    for (uint64_t b0 = (uint64_t)blockIdx.x * SLOW_TPB; b0 < n_def; b0 += (uint64_t)gridDim.x * SLOW_TPB) {
        const bool have = b0 + lane < n_def;
        uint64_t s = 0, e = 0;
        if (have) {
            s = a.deferred[b0 + lane];
            e = ~0ull;
            for (uint64_t p = s & ~15ull; e == ~0ull; p += 16) {     // the buffer is 16-byte aligned and zero padded far beyond n_bytes
                const uint4 v = *(const uint4 *)(a.gaf + p);
                uint32_t m = eq_mask16(v, 0x0A0A0A0Au) | eq_mask16(v, 0x0D0D0D0Du);
                if (p + 16 > a.n_bytes) m |= p >= a.n_bytes ? 0xFFFFu : (0xFFFFu << (a.n_bytes - p)) & 0xFFFFu;   // the text ends here
                if (p < s) m &= 0xFFFFu << (s - p);
                if (m) e = p + (uint64_t)__builtin_ctz(m);
            }
        }
        const uint64_t a0 = s & ~15ull;
        const uint64_t span = have ? ((e - a0 + 15) & ~15ull) : 0;        // bytes of the aligned blocks that hold the line
        const uint32_t want = span <= SLOW_MAXLINE ? (uint32_t)span : 0u;
        uint32_t tot;
        const uint32_t off = wave_excl_scan(want, tot);
        const bool staged = have && want && off + want <= SLOW_LDS;
        if (staged)
            for (uint32_t o = 0; o < want; o += 16) *(uint4 *)(stage + off + o) = *(const uint4 *)(a.gaf + a0 + o);
        __syncthreads();
        if (have) {
            const uint8_t *t = staged ? (const uint8_t *)stage + off : a.gaf;
            const uint64_t s2 = staged ? s - a0 : s;
            SlowEmit em{&a, a.base_offset + s};
            int rc = slow_line(a.g, t, s2, s2 + (e - s), em);
            if (rc) atomicMin(&a.st->err, ((a.base_offset + s) << 3) | (unsigned long long)rc);
        }
        __syncthreads();
    }
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

__global__ __launch_bounds__(TPB) void k_geno_maxn(GenoArgs a) {
    uint64_t r = (uint64_t)blockIdx.x * TPB + threadIdx.x;
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

__global__ __launch_bounds__(TPB) void k_genotype(GenoArgs a) {
    uint64_t r = (uint64_t)blockIdx.x * TPB + threadIdx.x;
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
