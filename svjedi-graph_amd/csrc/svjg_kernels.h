// HIP kernels of libsvjg_hip.so (gfx950 / MI355X, wave64).  No MFMA anywhere: this is byte / integer
// work bounded by HBM reads.
//
//   k_classify_main  persistent single-wave workers over 8 KB stripes of GAF text (see the comment in front of the kernel):
//                      coalesced HBM -> register -> LDS staging, byte classes by bit planes (svjg_planes.h), rank-indexed
//                      lists of line starts / orientation marks, then loop-free per-line, per-node and per-link phases with
//                      a perfect-hash node-record table, packed 64-bit (ref | alt << 32) atomics into the per-SV count vector
//   k_classify_slow  one lane per deferred line, exact string path (svjg::slow_line) on an LDS copy of the line
//   k_logfact_*      log10(i!) table in double-double for the binomial term
//   k_genotype       one VCF row per lane, fp64 / double-double likelihoods (predict-genotype.py:281-325)
#pragma once
#include <hip/hip_runtime.h>
#define SVJG_TAB_AS __attribute__((address_space(3)))      // the exact routine's per-node scratch and piece tables are LDS arrays here (svjg_line.h)
#include "svjg_line.h"
#include "svjg_planes.h"
#include "svjg_pass.h"

namespace svjg {

constexpr uint32_t WG = 64;                      // classify kernel: ONE wave per workgroup = one autonomous worker (no workgroup barriers anywhere)
constexpr uint32_t TPB = 256;                    // block size of the small per-row / per-line kernels
constexpr uint32_t PIECES = 4;                   // 16-byte pieces of text per lane and half
constexpr uint32_t SPAN = PIECES * 16;           // bytes of byte classification per lane and half: one 64-bit mask per byte class
constexpr uint32_t HALF = SPAN * WG;             // 4 KB: what the 64 lanes classify at once
constexpr uint32_t NHALF = 2;
constexpr uint32_t TEXT = NHALF * HALF;          // 8 KB staged in LDS per stripe
// A worker owns the lines that START in its region of the text and walks them in stripes: a stripe stages TEXT bytes from the
// 16-byte block that holds the first line not yet worked off, handles every line that ends inside the staged text, and the
// next stripe begins at the first line that did not (so no byte is classified twice except the tail of one line per stripe,
// and a line of up to ~8 KB never needs a second launch).  Phase B turns the staged text into rank-indexed lists (positions
// are offsets into the staged text):
// The lists are sized for ordinary line shapes (a 12-column line with a handful of tags is >= ~100 bytes); a stripe whose text
// is denser than that is first cut down to its first half and only then sent to the exact path as a whole.
constexpr uint32_t MAXL = 64;                       // line starts per stripe = lines of one round (one line per lane in the line phase)
#ifndef SVJG_CAP_O
#ifdef SVJG_W16
#define SVJG_CAP_O 208
#else
#define SVJG_CAP_O 216
#endif
#endif
constexpr uint32_t CAP_O = SVJG_CAP_O;              // orientation marks ('<' '>') per stripe
constexpr uint32_t KMAX = 64;                    // path nodes of one node pass (one wave, one node per lane)
// A line of more than KMAX nodes — up to KLONG, what the stripe's list of marks and the 8-bit node count of a line's record hold — is
// walked in sub-passes of 64 nodes that overlap by one (the step into the next sub-pass's first node is that sub-pass's own), twice:
// once to learn the path's total length (the right-hand overlap test needs it, filter-alignments.py:271) and that its ids rise or
// fall all the way or, where they turn, whether a name comes twice, once to count.  Where no name comes twice list.index is the position
// itself; where one does, sweep 1 takes every node's first occurrence over the WHOLE line (the ids of all nodes wait in LDS, the running
// path lengths in a few words of global memory of the worker's own).
#ifdef SVJG_W16
constexpr uint32_t KLONG = KMAX;                 // (measurement variant: the per-line records live where a long line's ids would wait)
#else
constexpr uint32_t KLONG = CAP_O < 255u ? CAP_O : 255u;
#endif
constexpr uint32_t LONG_LOG = 256;               // per worker, in global memory: [0, LONG_LOG) the running path length behind every node of a long line,
constexpr uint32_t LONG_WORDS = 3 * LONG_LOG;    // [LONG_LOG, 3 * LONG_LOG) two words per link: the hits a first sweep has found and holds back (ClassifyArgs::long_pre)
static_assert(KLONG <= LONG_LOG && KLONG * 4 <= TEXT / 8 + 16, "a long line's per-node words fit their places");
constexpr uint32_t LRW = MAXL;                   // lines per round
static_assert(TEXT + 1 < 65535, "text offsets are kept in 16 bits, 0xFFFF = none");

// LDS of one worker (bytes); the hardware hands LDS out in units of 1280 bytes
constexpr uint32_t L_TEXT = 0;                                             // staged text + slack for the word reads behind a name / column
#ifdef SVJG_W16
// MEASUREMENT VARIANT (tools/mkvariant.sh w16 -DSVJG_W16): sixteen workers per CU.  No non-digit bitmap (the nine decimal columns are
// NOT tested for digits: results are right only for well-formed text), the per-line records take the tab bitmap's place after the
// line phase, the marks' list is 16 bits of position + 8 bits of line per mark, 208 marks per stripe.
constexpr uint32_t L_TBM = L_TEXT + TEXT + 64;
constexpr uint32_t L_NDBM = L_TBM;                                         // (no such bitmap)
constexpr uint32_t L_RL = L_TBM;
constexpr uint32_t L_OPL = L_TBM + TEXT / 8 + 16;                          // u16[CAP_O + 8] positions, then u8[CAP_O + 8] line + 1
constexpr uint32_t L_LINE = L_OPL + (CAP_O + 8) * 3;
constexpr uint32_t LDS_MAIN = L_LINE + (MAXL + 8) * 4;
constexpr uint32_t LDS_GRANULE = 1280;
static_assert(LDS_MAIN <= 8 * LDS_GRANULE, "sixteen workers per CU");
static_assert(L_TBM % 16 == 0 && L_OPL % 16 == 0 && L_LINE % 4 == 0, "LDS alignment");
#else
constexpr uint32_t L_NDBM = L_TEXT + TEXT + 64;                            // u32[TEXT/32 + 4] one bit per byte: neither a digit nor a tab (line phase only)
constexpr uint32_t L_RL = L_NDBM;                                          // uint4[LRW] per line, written at the END of the line phase (the bitmap is dead by then):
                                                                           //   need_l, need_r, first mark (rel.) | k << 16 | status << 24, tab behind the path
constexpr uint32_t L_TBM = L_NDBM + TEXT / 8 + 16;                         // u32[TEXT/32 + 4] one bit per byte: a tab (the line phase walks a line's columns on it)
constexpr uint32_t L_OPL = L_TBM + TEXT / 8 + 16;                          // u32[CAP_O + 8]   orientation mark #o: position | line (ordinal in the stripe) << 16
constexpr uint32_t L_LINE = L_OPL + (CAP_O + 8) * 4;                       // u16[2][MAXL + 8] line #l: start, marks in front of it ; [n].start = 0xFFFF
constexpr uint32_t LDS_MAIN = L_LINE + (MAXL + 8) * 4;
constexpr uint32_t LDS_GRANULE = 1280;
static_assert(LRW * 16 <= TEXT / 8 + 16, "the per-line records fit the bitmap they replace");
static_assert(LDS_MAIN <= 9 * LDS_GRANULE, "fourteen workers per CU");
static_assert(L_NDBM % 16 == 0 && L_TBM % 16 == 0 && L_OPL % 16 == 0 && L_LINE % 16 == 0 && L_RL % 16 == 0, "LDS alignment");
#endif

// status words (device)
struct DevStatus {
    unsigned long long n_lines;
    unsigned long long n_deferred;       // entries appended to the deferred list
    unsigned long long n_recs;           // hit records appended
    unsigned long long err;              // min over (file offset << 3 | exception class); ~0 = none
    unsigned int non_ascii;
    unsigned int overflow;               // bit 0: deferred list, bit 1: hit-record buffer, bit 2: list of lines for the host
    unsigned long long next_chunk;       // k_classify_main: small chunks handed out so far (zero at launch)
    unsigned long long n_host;           // lines set aside for the host (SVJG_EXC_ASK_HOST)
    unsigned long long cause[8];         // deferred lines by cause (DC_*)
};

struct ClassifyArgs {
    const uint8_t *gaf;                  // resident text, allocation padded with >= TEXT + 64 zero bytes
    uint64_t begin;                      // first byte to classify (a line start); what lies in front of it belongs to another launch
    uint64_t n_bytes;                    // end of the text to classify; >= TEXT + 64 zero bytes follow it in the buffer
    uint64_t base_offset;
    GraphView g;                         // global-memory views
    uint32_t all_slow;
    uint32_t want_hits;
    uint64_t region;                     // bytes of text of a worker's FIRST chunk (multiple of 16): worker w owns the lines that start in [w * region, (w + 1) * region)
    uint64_t small;                      // the text behind grid * region goes in chunks of this many bytes (multiple of 16) to whoever is free next; 0: there is none
    uint32_t diag;                       // measurement only (SVJG_DIAG): 1 stop after B, 2 stop after R1, 8 no atomics
    unsigned long long *counts;          // [n_slots] ref | alt << 32
    uint64_t *deferred;  uint64_t deferred_cap;
    svjg_hitrec *recs;   uint64_t rec_cap;
    uint64_t *host_lines; uint64_t host_cap;
    DevStatus *st;
    unsigned long long *dbg;             // measurement only (SVJG_DIAG & 16): per-phase cycle sums of lane 0 of every worker
    uint32_t *long_pre;                  // LONG_WORDS words per worker: running path length behind every node of a long line (written in sweep 0, read in sweep 1 by a line that comes back to a node), and the hits of its links, held back until the line's last name is known
};

// (the exact-path kernels look for a line's end with these)
__device__ inline uint32_t zero_bytes(uint32_t t) {                    // 0x80 in every byte of t that is zero (exact)
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}
__device__ inline uint32_t eq_mask16(uint4 v, uint32_t pat) {          // 16-bit mask of bytes equal to pat's byte
    // byte flags (0x80 / 0x00) gathered with two chained v_dot4_u32_u8 per half: weights 1,2,4,8 / 16,32,64,128
    const uint32_t lo = __builtin_amdgcn_udot4(zero_bytes(v.x ^ pat), 0x08040201u, __builtin_amdgcn_udot4(zero_bytes(v.y ^ pat), 0x80402010u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(zero_bytes(v.z ^ pat), 0x08040201u, __builtin_amdgcn_udot4(zero_bytes(v.w ^ pat), 0x80402010u, 0u, false), false);
    return (lo | (hi << 8)) >> 7;
}

// Wave prefix sums without LDS traffic: DPP row shifts inside the four rows of 16 lanes, then the row totals are
// broadcast into the following rows (row_bcast:15 / row_bcast:31): six v_add_u32_dpp.  Written out: from the builtins the compiler
// makes three instructions per step (clear, v_mov_b32_dpp, add).  (s_nop: a DPP operand written by the instruction before needs
// two wait states on gfx9, and the hazard pass does not look into inline assembly.)
__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    uint32_t x = v;
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(x));
    return x;
}
// two of them, interleaved (each fills one of the other's wait states)
__device__ inline void wave_incl_scan2(uint32_t &x, uint32_t &y) {
#define SVJG_STEP(ctl) "s_nop 0\n\tv_add_u32_dpp %0, %0, %0 " ctl "\n\tv_add_u32_dpp %1, %1, %1 " ctl "\n\t"
    asm("s_nop 0\n\t"
        SVJG_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") SVJG_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SVJG_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0") SVJG_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SVJG_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") SVJG_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(x), "+v"(y));
#undef SVJG_STEP
}
// ballot without the bool -> int detour of __ballot (one v_cmp into an SGPR pair), value of a lane whose number is wave-uniform
__device__ inline unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Lane predicates as wave masks.  A ballot of a COMPOUND condition (a && b) costs two extra vector instructions with this compiler
// (the and of two compares lives in an SGPR pair, is turned into 0 / 1 per lane and compared again); a compare that yields its mask
// at once (llvm.amdgcn.icmp), scalar logic on the masks, and a mask taken as a lane predicate where one is needed cost none.
typedef unsigned long long wmask;
#define RARELY(c) __builtin_expect((c) != 0, 0)                     /* a wave-uniform branch the usual text does not take: its block is laid out of line, so the usual path has no taken branch there */
__device__ inline wmask m_eq(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 32); }
__device__ inline wmask m_ne(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 33); }
__device__ inline wmask m_gt(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 34); }
__device__ inline wmask m_ge(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 35); }
__device__ inline wmask m_lt(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 36); }
__device__ inline wmask m_le(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 37); }
__device__ inline bool in_mask(wmask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }   // this lane's bit of a wave-uniform mask
__device__ inline uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// value of the lane below, lane 0 gets `first`
__device__ inline uint32_t lane_below_or(uint32_t v, uint32_t first) { return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xF, 0xF, false); }   // wave_shr:1
// value of the lane below / above (lane 0 / lane 63 keep their own), one DPP move instead of a trip through the LDS crossbar
__device__ inline uint32_t lane_below(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false); }   // wave_shr:1
__device__ inline uint32_t lane_above(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xF, 0xF, false); }   // wave_shl:1
__device__ inline uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
    const uint32_t x = wave_incl_scan(v);
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}

enum : uint32_t { ST_NONE = 0, ST_OK = 1, ST_NOHIT = 2, ST_DEFER = 3 };   // per-line status inside a round; ST_DEFER + cause: why the line takes the exact path
enum : uint32_t { DC_COLUMNS = 0,      // the line's columns are not twelve plain ones (blanks, signs, too few, Alen = 0, marks outside the path column, ...)
                  DC_IDF = 1,          // a 64-byte span of the line holds the pair "d:" (an id:f: tag, or a false alarm)
                  DC_LONG_PATH = 2,    // more than KMAX path nodes
                  DC_NAME = 3,         // a node name the name table does not hold (unknown, hazard-prone, longer than 64 bytes, alt length unknown), a path of 2^32 bp
                  DC_STRIPE = 4,       // the whole stripe: denser than the lists hold, a line longer than the staged text, or the caller asked for the exact path
                  DC_N = 5 };

// Path segment text[a0, a0+L), 1 <= L <= 48: its window words (svjg_line.h: name_windows) and the 64-bit pre-hash of the node-name
// table (name_prehash), in three parts: words 0..5 (three 8-byte windows that cover a name of up to 24 bytes; d[6..7] = 0), words 6
// and 7 (names of 25..32 bytes) and words 8..11 (33..48 bytes: contig names like chr1_KI270706v1_random), each with what it adds to
// the hash.  A window ends where the name ends at the latest, so the words are read from the staged text as they are (LDS words at
// any byte address: tools/ubench/lds_unaligned.hip); only a name of fewer than eight bytes has foreign bytes in its windows, and
// `any_short` (wave-uniform) says whether some lane of the pass holds one.
typedef uint32_t u32_any __attribute__((aligned(1)));
typedef unsigned long long u64_any __attribute__((aligned(1)));
__device__ inline uint64_t name_words_head(const uint8_t *text, uint32_t a0, uint32_t L, bool any_short, uint32_t d[8]) {
    const uint32_t o2 = (uint32_t)min(max((int32_t)L - 8, 0), 16), o1 = o2 >> 1;   // (svjg_line.h: name_windows)
    const uint8_t *p = text + a0;
    unsigned long long w0 = *(const u64_any *)p, w1 = *(const u64_any *)(p + o1), w2 = *(const u64_any *)(p + o2);
    if (RARELY(any_short)) {                                                  // (o1 = o2 = 0 for these lanes: all three windows are the first eight bytes)
        const unsigned long long m = L < 8u ? ~(~0ull << (8u * L)) : ~0ull;
        w0 &= m; w1 &= m; w2 &= m;
    }
    d[0] = (uint32_t)w0; d[1] = (uint32_t)(w0 >> 32); d[2] = (uint32_t)w1; d[3] = (uint32_t)(w1 >> 32); d[4] = (uint32_t)w2; d[5] = (uint32_t)(w2 >> 32);
    d[6] = 0u; d[7] = 0u;
    const uint32_t C[6] = {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du};
    uint64_t h = (uint64_t)L * 0x7FEB352Du;
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i) h += (uint64_t)d[i] * C[i];
    return h;
}
// (L is a probed lane's length — 1..48 — or 8 for the others: the addresses stay inside the staged text)
__device__ inline uint64_t name_words_tail(const uint8_t *text, uint32_t a0, uint32_t L, uint32_t d[8]) {
    const unsigned long long w = *(const u64_any *)(text + a0 + (L > 24u ? L - 8u : 0u));
    d[6] = L > 24u ? (uint32_t)w : 0u;
    d[7] = L > 24u ? (uint32_t)(w >> 32) : 0u;
    return (uint64_t)d[6] * 0xFD7046C5u + (uint64_t)d[7] * 0xB55A4F09u;
}
// words 8..11 (names of 33..48 bytes) are not kept in registers while the record travels: they are read from the staged text twice,
// for the hash and — in the branch only a pass with such a name takes — for the compare
// (r06: a name of 33..40 bytes has two of them — its bytes [L - 16, L - 8) —, words 10 and 11 are zero: svjg_line.h: name_windows)
__device__ inline void name_words_far(const uint8_t *text, uint32_t a0, uint32_t L, uint32_t f[4]) {
    const u32_any *w = (const u32_any *)(text + a0 + (L > 40u ? L - 24u : L > 32u ? L - 16u : 0u));
    f[0] = L > 32u ? w[0] : 0u; f[1] = L > 32u ? w[1] : 0u; f[2] = L > 40u ? w[2] : 0u; f[3] = L > 40u ? w[3] : 0u;
}
// the first L - 48 bytes of a name of 49..64 bytes as four words, zero behind them (svjg_line.h: name_prefix_words), from the staged text
__device__ inline void name_prefix_lds(const uint8_t *text, uint32_t a0, uint32_t L, uint32_t p[NAME_PFX_WORDS]) {
    const u32_any *w = (const u32_any *)(text + a0);
    const int32_t n = (int32_t)L - (int32_t)(4u * NAME_WORDS);
#pragma unroll
    for (int32_t i = 0; i < (int32_t)NAME_PFX_WORDS; ++i) {
        const int32_t k = n - 4 * i;                                      // bytes of word i that belong to the prefix
        p[i] = k >= 4 ? w[i] : k <= 0 ? 0u : (w[i] & ((1u << (8 * k)) - 1u));
    }
}
__device__ inline uint64_t name_words_tail2(const uint8_t *text, uint32_t a0, uint32_t L) {
    uint32_t f[4];
    name_words_far(text, a0, L, f);
    return (uint64_t)f[0] * 0x94D049BBu + (uint64_t)f[1] * 0xBF58476Du + (uint64_t)f[2] * 0x2545F491u + (uint64_t)f[3] * 0x9FB21C65u;
}

// record of the node-name table (svjg_host_tables.h): r0 = window words 0..3, r1 = words 4, 5 | meta | length in bp,
// r2.xy = words 6, 7 (only names longer than 24 bytes look at them); words 8..11 (r2.zw, r3.xy) are compared by the caller in
// the branch only a pass with such a name takes
__device__ inline bool name_match(const uint4 r0, const uint4 r1, const uint4 r2, const uint32_t d[8], uint32_t L) {
    // one OR of differences instead of a chain of compares (three-input bit operations: (a ^ b) | c is one instruction)
    uint32_t diff = ((r1.z & NAME_LEN_MASK) ^ (L - 1u)) | (r0.x ^ d[0]) | (r0.y ^ d[1]) | (r0.z ^ d[2]) | (r0.w ^ d[3]) | (r1.x ^ d[4]) | (r1.y ^ d[5]);
    const uint32_t tail = (r2.x ^ d[6]) | (r2.y ^ d[7]);
    diff |= L > 24u ? tail : 0u;
    return diff == 0u;
}

// LDS traffic between lanes of ONE wave: DS operations of a wave execute in order, the fences only pin the compiler
// A worker's words in global memory (ClassifyArgs::long_pre) are written by some lanes of the wave and read later by others: worker = wave =
// workgroup, so workgroup scope orders them (the stores have left the wave before the loads are issued; both meet in the CU's L1 / the L2).
// (Agent-scope fences write the XCD's L2 back and invalidate it: +3.5 ms a launch on the long-read block, measured.)
__device__ inline void long_words_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The classify kernel.  ONE WAVE = one autonomous worker (64-lane workgroups, no workgroup barrier anywhere): it owns the lines
// that start in its region of the text and walks them in stripes of TEXT = 8 KB staged in LDS.  A stripe begins at the 16-byte
// block that holds the first line not yet worked off; lines that end inside the staged text are handled, the first one that
// does not is where the next stripe begins (a line longer than a stripe goes to the exact path).  The next stripe's HBM loads
// are issued into registers as soon as its start is known (right after the byte classes), so the only HBM read of the text
// overlaps the line and node phases.  Waves of a CU drift apart, so the dense byte classification of one fills the issue
// slots another leaves while it waits for LDS / table round trips.  Per stripe:
//   A   registers -> LDS (16 B per lane and piece, coalesced on the HBM side)
//   B1  two 64-byte SPANs per lane: the span's sixteen words are transposed into eight bit planes (svjg_planes.h) and every byte
//       class is a boolean function of the planes: 64-bit masks of line terminators and orientation marks ('<' '>') in registers,
//       the bitmaps "tab" and "neither digit nor tab" -> LDS; also "any byte >= 0x80" and the "id:f:" filter (the pair "d:");
//       wave prefix sums of the counts
//   B2  every terminator / mark knows its ordinal in the stripe: rank-indexed lists -> LDS
//         LINE[l] = start of line l, marks in front of it
//         OPL[o]  = position of mark o | line that holds it << 16
//   then in rounds of up to 64 lines
//   (no loops over bytes or bits from here on: a line's columns are found on the tab bitmap, its j-th node starts at OPL[marks in front of the line + j]):
//   R1 one LINE per lane: twelve column boundaries, column lengths, digits-only test on the bitmap, the decimal
//      values that matter (Tlen, Ts, Te; Alen only as "zero or not"), path geometry
//   NP node passes over up to 64 consecutive marks covering whole lines, one path NODE per lane, everything in registers:
//      name -> the bucket's displacement -> the ONE record the name can be in (perfect hash); then id / length, running path
//      length by a wave scan, first occurrence of every name (the reference's list.index / str.split
//      quirks) by DPP wave shifts, overlap test, link among the record's inline links (link table on a miss), one 64-bit
//      atomic (ref | alt << 32) per hit, optional hit records
//   R6 deferred-line offsets, one aggregated atomic per wave

// Phase B1 for one lane: classes of the SPAN bytes at text + slot * SPAN (slot = half * 64 + lane), by bit planes (svjg_planes.h).
// terminator = '\n', or a '\r' not followed by '\n' (Python universal newlines).
struct SpanFlags {
    uint32_t high;          // some byte >= 0x80 (per lane)
    uint32_t idf;           // the byte pair "d:" begins in this lane's spans (per lane)
    uint32_t dee_last;      // half 0: the span's last byte is 'd' (the ':' would be the next lane's, or the next half's, first byte)
    uint32_t idsum;         // per half one byte: pairs "d:" whose 'd' lies in this lane's span — min(count, 3) << 6 | position of the last one
};
__device__ inline uint32_t idsum_of(uint32_t pair_lo, uint32_t pair_hi) {   // (pair_lo | pair_hi != 0)
    const uint32_t cnt = (uint32_t)__popc(pair_lo) + (uint32_t)__popc(pair_hi);
    const uint32_t pos = pair_hi ? 63u - (uint32_t)__builtin_clz(pair_hi) : 31u - (uint32_t)__builtin_clz(pair_lo);
    return ((cnt < 3u ? cnt : 3u) << 6) | pos;
}
__device__ inline uint32_t lane_above_or0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }   // wave_shl:1, lane 63 gets 0
__device__ inline void classify_span(const ClassifyArgs &a, const uint8_t *text, uint32_t *ndbm, uint32_t *tbm, uint32_t half, uint32_t slot, uint64_t c0, uint32_t V,
                                     unsigned long long &NL, unsigned long long &ORI, SpanFlags &fl) {
    const uint32_t sp = slot * SPAN;
    // rotated piece order: the 16-byte LDS reads of a wave hit different banks.  The planes are those of the span rotated by
    // 16 * rot bytes, so every 64-bit mask is rotated back at the end: a halfword permutation, two v_perm_b32.
    const uint32_t rot = (slot >> 2) & 3u;
    uint32_t w[16];
#pragma unroll
    for (uint32_t c = 0; c < PIECES; ++c) {
        const uint4 v = *(const uint4 *)(text + sp + ((c + rot) & 3u) * 16);
        w[c * 4 + 0] = v.x; w[c * 4 + 1] = v.y; w[c * 4 + 2] = v.z; w[c * 4 + 3] = v.w;
    }
    span_planes(w);
    const HalfClasses lo = half_classes(w), hi = half_classes(w + 8);
    const uint32_t sel_lo = (0x0B0A0908u - 0x02020202u * rot) & 0x07070707u, sel_hi = sel_lo ^ 0x04040404u;   // halfword t of the result = halfword (t - rot) & 3
    auto place = [&](uint32_t l, uint32_t h) -> unsigned long long {
        return (unsigned long long)perm_b32(h, l, sel_lo) | ((unsigned long long)perm_b32(h, l, sel_hi) << 32);
    };
    unsigned long long nl64 = place(lo.nl, hi.nl);
    const unsigned long long tab64 = place(lo.tab, hi.tab), ori64 = place(lo.ori, hi.ori), nd64 = place(lo.nd, hi.nd);
    fl.high |= lo.high | hi.high;
    // "d:" — the pair may straddle the span's end: the next lane's first byte comes by DPP, the next half's is looked at by the caller
    {
        const unsigned long long dee = place(lo.dee, hi.dee), col = place(lo.colon, hi.colon);
        const uint32_t nxt = lane_above_or0((uint32_t)col);
        const uint32_t s_lo = __builtin_amdgcn_alignbit((uint32_t)(col >> 32), (uint32_t)col, 1), s_hi = __builtin_amdgcn_alignbit(nxt, (uint32_t)(col >> 32), 1);
        const uint32_t pair_lo = (uint32_t)dee & s_lo, pair_hi = (uint32_t)(dee >> 32) & s_hi;
        fl.idf |= pair_lo | pair_hi;
        if (RARELY(m_ne(pair_lo | pair_hi, 0u))) {                       // (wave-uniform: no tag minigraph writes holds the pair)
            if ((pair_lo | pair_hi) != 0u) fl.idsum |= idsum_of(pair_lo, pair_hi) << (half ? 8 : 0);
        }
        fl.dee_last = (uint32_t)(dee >> 63);
    }
    // carriage returns (no text file has them in practice)
    if ((lo.cr | hi.cr) != 0) {
        unsigned long long cr = place(lo.cr, hi.cr);
        while (cr) {
            const uint32_t b = (uint32_t)__builtin_ctzll(cr); cr &= cr - 1;
            const uint32_t q = sp + b;
            const uint32_t staged = (slot / WG + 1u) * HALF;              // (the half behind this one may not be in LDS yet)
            const uint8_t nx = (q + 1 < staged) ? text[q + 1] : ((c0 + q + 1 < a.n_bytes) ? a.gaf[c0 + q + 1] : 0);
            if (nx != '\n') nl64 |= 1ull << b;
        }
    }
    if (sp + SPAN > V) nl64 &= sp >= V ? 0ull : ((1ull << (V - sp)) - 1ull);   // ignore anything at or beyond the valid length
#ifndef SVJG_W16
    *(uint2 *)(ndbm + slot * 2) = make_uint2((uint32_t)nd64, (uint32_t)(nd64 >> 32));
#endif
    *(uint2 *)(tbm + slot * 2) = make_uint2((uint32_t)tab64, (uint32_t)(tab64 >> 32));
    NL = nl64; ORI = ori64;
}

// decimal column text[a, a + n), 1 <= n <= 9, known to hold digits only -> value.  Straight-line SWAR on its first eight
// bytes: digits folded pairwise (no per-digit loop, no divergence).
__device__ inline uint32_t field_val(const uint8_t *text, uint32_t a, uint32_t n) {
    const u32_any *w = (const u32_any *)(text + a);
    const uint32_t lo0 = w[0] & 0x0F0F0F0Fu;                              // chars a .. a+3 as digit values
    const uint32_t hi0 = w[1] & 0x0F0F0F0Fu;                              // chars a+4 .. a+7
    const uint32_t n8 = n < 8 ? n : 8;
    // the first n8 bytes move to the top of the 64-bit (hi:lo): what follows the column drops out, leading bytes are zero digits
    const unsigned long long x = (((unsigned long long)hi0 << 32) | lo0) << ((8 * (8 - n8)) & 63u);
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    // digit pairs by shift-and-add (x * 10 = (x << 3) + (x << 1)), the rest with 24-bit multiplies
    const uint32_t pl = (((lo << 3) + (lo << 1)) + (lo >> 8)) & 0x00FF00FFu, ph = (((hi << 3) + (hi << 1)) + (hi >> 8)) & 0x00FF00FFu;
    uint32_t r = __umul24(__umul24(pl & 0xFFu, 100u) + (pl >> 16), 10000u) + __umul24(ph & 0xFFu, 100u) + (ph >> 16);
    if (n == 9) r = r * 10u + ((uint32_t)text[a + 8] & 0xFu);
    return r;
}

// the same column is zero (every digit a '0')
__device__ inline bool field_is_zero(const uint8_t *text, uint32_t a, uint32_t n) {
    const u32_any *w = (const u32_any *)(text + a);
    const uint32_t lo0 = w[0] & 0x0F0F0F0Fu, hi0 = w[1] & 0x0F0F0F0Fu;
    const uint32_t n8 = n < 8 ? n : 8;
    const unsigned long long x = (((unsigned long long)hi0 << 32) | lo0) << ((8 * (8 - n8)) & 63u);
    return x == 0 && (n < 9 || ((uint32_t)text[a + 8] & 0xFu) == 0);
}

// no bit set in bits [a, a + n) of a bitmap, n <= 64 (three words are read: the bitmap is padded)
__device__ inline bool bits_clear(const uint32_t *bm, uint32_t a, uint32_t n) {
    const uint32_t *w = bm + (a >> 5);
    const uint32_t sh = a & 31u;
    const unsigned long long lo = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
    unsigned long long x = lo >> sh;
    if (sh) x |= (unsigned long long)w[2] << (64u - sh);
    const unsigned long long m = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    return (x & m) == 0;
}

// Columns of a line on the tab bitmap (one bit per byte of the staged text).  tab_window: the 32 bits from position p on;
// take_tab: the first tab of a window whose bit 0 is position p -> its position (TEXT = none) and the window without it.
// Three decimal columns of at most nine characters and their tabs fit one window.  tab_from: first tab in [p, lim), any distance.
// (p <= TEXT + 1: the bitmap has two words of slack, kept zero.)
__device__ inline uint32_t tab_window(const uint32_t *tbm, uint32_t p) {
    const uint32_t *w = tbm + (p >> 5);
    return __builtin_amdgcn_alignbit(w[1], w[0], p & 31u);
}
__device__ inline uint32_t take_tab(uint32_t &win, uint32_t p) {
    const uint32_t t = win ? p + (uint32_t)__builtin_ctz(win) : TEXT;
    win &= win - 1u;
    return t;
}
__device__ inline uint32_t tab_near(const uint32_t *tbm, uint32_t p) {     // first tab in [p, p + 32), or TEXT
    const uint32_t win = tab_window(tbm, p);
    return win ? p + (uint32_t)__builtin_ctz(win) : TEXT;
}
__device__ inline uint32_t tab_from(const uint32_t *tbm, uint32_t p, uint32_t lim) {
    for (uint32_t q = p; q < lim; q += 32) {
        const uint32_t win = tab_window(tbm, q);
        if (win) { const uint32_t t = q + (uint32_t)__builtin_ctz(win); return t < lim ? t : TEXT; }
    }
    return TEXT;
}

// "id:f:" anywhere in a line changes what the reference does with it (filter-alignments.py:193-196: float() of the tag value may
// raise, Alen == 0 no longer does): such lines take the exact path.  The filter is the byte pair "d:" (classify_span): no tag
// minigraph writes and hardly any read name holds it, and a false alarm only costs the stripe's lines the exact path.

// Static wave priorities (s_setprio): the byte classification of phases A / B1 is dense VALU work, the line and node phases
// are chains of LDS and memory round trips.  A wave in the latency-bound phases wins the issue arbitration against the
// waves that classify bytes, so its loads go out early and the classifying waves fill the gaps.
#ifndef SVJG_P_B2
#define SVJG_P_B2 1          /* list building */
#define SVJG_P_R1 2          /* line phase */
#define SVJG_P_LOAD 3        /* node pass until its table loads are issued */
#define SVJG_P_REST 2        /* rest of the node pass */
#endif
#ifndef SVJG_P_A
#define SVJG_P_A 2          /* registers -> LDS (waits for the prefetch: high, so that the next dense phase starts early; measured) */
#endif
constexpr int P_A = SVJG_P_A;
constexpr int P_B2 = SVJG_P_B2, P_R1 = SVJG_P_R1, P_LOAD = SVJG_P_LOAD, P_REST = SVJG_P_REST;
// the ablation knobs (ClassifyArgs::diag) exist only in -DSVJG_ABLATE builds (tools/ablate.sh): the shipped kernel does not test them
#ifdef SVJG_ABLATE
#define DIAG(bit) ((a.diag & (bit)) != 0)
#else
#define DIAG(bit) false
#endif
#ifndef SVJG_MINW
#define SVJG_MINW 4          /* four waves per SIMD: at most 128 VGPRs (fourteen workers per CU by LDS) */
#endif
__device__ inline unsigned long long low_bits64(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }
__device__ inline uint32_t clamp64(uint32_t hi, uint32_t lo) { return hi > lo ? (hi - lo < 64u ? hi - lo : 64u) : 0u; }   // min(max(hi - lo, 0), 64)

// r06 — may a node come twice in a line of K > 64 nodes (K <= KLONG)?  ids[0, K) = the line's node ids (low 24 bits) in path order, in the
// words of LDS where the tab bitmap was; the words behind them, up to word 256, are free: a bitmap of B = 32 * nb bits there (nb = 128 / 64 / 32
// words for K <= 128 / 192 / 216), bit (id mod M) per node, set with an LDS atomic OR that returns what was there.  No bit met twice => no id
// twice (exact).  A bit met twice => maybe: two ids a multiple of M apart look alike.  The ids follow the genome, so the nodes of one stretch of
// a path fall on neighbouring bits (an inversion walks its stretch backwards: still distinct bits), and only stretches on different contigs —
// a line that follows a breakend — can fall on each other, which with a few such stretches happens for about a third of those lines.  So the
// question is asked again with another modulus (B, then 15/16, 14/16 ... 9/16 of it: the stretches then lie differently) — a line that really
// comes back to a node is caught by every one of them, a line that does not is let go by the first that tells its stretches apart — and only
// what all eight flag takes the caller's full search, which is exact.  One call per long line, ~30 instructions a try, two tries on average.
__device__ inline bool long_line_may_repeat(uint32_t *ids, uint32_t K, uint32_t lane) {
    const uint32_t nb = K <= 128u ? 128u : K <= 192u ? 64u : 32u, B = nb * 32u;
    uint32_t *bm = ids + 256u - nb;
    for (uint32_t t = 0; t < 8u; ++t) {
        const uint32_t M = B - t * (B >> 4) - (t ? 2u * t + 1u : 0u);        // 2048, 1917, 1787, 1657, 1527, 1397, 1267, 1137 (B = 2048)
        const uint32_t inv = 0xFFFFFFFFu / M;
        bm[lane & (nb - 1u)] = 0u;
        if (nb > 64u) bm[64u + lane] = 0u;
        wave_sync();
        uint32_t twice = 0u;
        for (uint32_t m = lane; m < K; m += WG) {
            const uint32_t x = ids[m] & 0x00FFFFFFu;
            uint32_t r = x - __umulhi(x, inv) * M;                            // x mod M (the quotient is at most one short)
            if (r >= M) r -= M;
            const uint32_t was = __hip_atomic_fetch_or(&bm[r >> 5], 1u << (r & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            twice |= (was >> (r & 31u)) & 1u;
        }
        if (ballot64(twice != 0u) == 0ull) return false;
        wave_sync();
    }
    return true;
}

__global__ __launch_bounds__(WG, SVJG_MINW) void k_classify_main(ClassifyArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *text = lds + L_TEXT;
    uint32_t *ndbm = (uint32_t *)(lds + L_NDBM);
    uint32_t *tbm = (uint32_t *)(lds + L_TBM);
#ifdef SVJG_W16
    uint16_t *OPLP = (uint16_t *)(lds + L_OPL);
    uint8_t *OPLL = lds + L_OPL + (CAP_O + 8) * 2;
#define OPL_PUT(j, v) do { const uint32_t v_ = (v); OPLP[j] = (uint16_t)v_; OPLL[j] = (uint8_t)((v_ >> 16) + 1u); } while (0)
#define OPL_POS(j) ((uint32_t)OPLP[j])
#else
    uint32_t *OPL = (uint32_t *)(lds + L_OPL);
#define OPL_PUT(j, v) OPL[j] = (v)
#define OPL_POS(j) (OPL[j] & 0xFFFFu)
#endif
    uint32_t *LINE = (uint32_t *)(lds + L_LINE);                        // start | marks in front << 16
    uint4 *RL = (uint4 *)(lds + L_RL);                                  // (the bitmap's bytes: see L_RL)

    const uint32_t lane = threadIdx.x;
    const GraphView g = a.g;

    // Work is handed out in chunks of text; a worker owns the lines that START in its chunk.  The first chunk of every worker is
    // fixed (most of an even share); the rest of the text goes in small chunks to whoever is free next: the fourteen workers of a CU
    // sit four, four, three and three on its SIMDs and those that share an issue port with three others are a quarter slower
    // (even shares: the last worker ended 15 % behind the average one).
    unsigned long long pos = a.begin + (unsigned long long)blockIdx.x * a.region;   // first byte not worked off yet (wave-uniform)
    if (pos >= a.n_bytes) return;
    unsigned long long rend = pos + a.region < a.n_bytes ? pos + a.region : a.n_bytes;   // lines starting before it belong to this chunk

    if (lane < 4) tbm[TEXT / 32 + lane] = 0;                            // slack of the tab bitmap (tab_near reads one word past a position)
    unsigned long long wave_lines = 0;
    // measurement only (build with -DSVJG_TIMING, run with SVJG_DIAG & 16): time this wave spends per phase
#ifdef SVJG_TIMING
    unsigned long long stamp = __builtin_readcyclecounter(), acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&](int ph) { const unsigned long long t = __builtin_readcyclecounter(); acc[ph] += t - stamp; stamp = t; };
    // (8 .. 10 split the node pass's load phase: these wait for the loads they stamp, which the shipped kernel does not do there)
#define tick_mem(ph) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tick(ph); } while (0)
#define tally(ph, n) do { acc[ph] += (n); } while (0)             // (5: nodes in passes, 10: passes, 11: sub-passes of long lines; of those 12: first sweep, counting, 13: first sweep, measuring only, 14: second sweep; 15: nodes in sub-passes)
#elif defined(SVJG_MARK)
    // static census (tools/isa): the phase boundaries show up as comments in the -S output
#define tick(ph) asm volatile("; MARK " #ph)
#else
    auto tick = [](int) {};
#endif
#ifndef SVJG_TIMING
#define tick_mem(ph) do { } while (0)
#define tally(ph, n) do { } while (0)
#endif

    // stripe prefetch registers: the FIRST HALF of a stripe waits in registers while the stripe in front of it is worked off (piece i
    // of the lane = bytes [(i * 64 + lane) * 16, + 16)); its second half is asked for when the first has gone to LDS and arrives
    // while the first half's bytes are classified — half the registers for the same overlap.
    constexpr uint32_t NPF = HALF / (WG * 16);
    uint4 pf[NPF];
    uint32_t pf_head = '\n';                                           // byte right before the stripe (decides whether it starts a line)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto fetch_half = [&](unsigned long long at) {                   // no bounds tests: the buffer is zero padded by TEXT + 64 bytes
        const uint4 *src = (const uint4 *)(a.gaf + at) + lane;
#pragma unroll
        for (uint32_t i = 0; i < NPF; ++i) {                             // non-temporal: the text is streamed once and should not displace the node records in L2
            const u32x4 v = __builtin_nontemporal_load((const u32x4 *)(src + i * WG));
            pf[i] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    // (the text begins at a line start whatever lies in front of it; later stripes take the byte in front of them from the staged
    //  text of the stripe before: a scalar load here would sit behind an s_waitcnt vmcnt(0) that also waits for the stripe's loads
    //  just issued — the whole HBM latency, every stripe)
  for (;;) {                                                             // chunks
    fetch_half(pos & ~15ull);
    { const unsigned long long c0 = pos & ~15ull; pf_head = c0 > a.begin ? a.gaf[c0 - 1] : (uint32_t)'\n'; }

    for (;;) {                                                           // stripes of a chunk
        const unsigned long long c0 = pos & ~15ull;
        const uint32_t own_lo = (uint32_t)(pos - c0);                   // lines starting in [own_lo, lim2) belong to this stripe
        const uint32_t V = (uint32_t)((a.n_bytes - c0 < (unsigned long long)TEXT) ? (a.n_bytes - c0) : (unsigned long long)TEXT);   // valid bytes staged
        const bool at_eof = c0 + V == a.n_bytes;

        // ---- A / B1, half by half: registers -> LDS, then the byte classes of the staged half (which also tell bytes >= 0x80 and
        //      the "id:f:" filter: every such tag holds the byte pair "d:"; a hit sends the stripe's lines to the exact path) ----------
        const uint32_t head_byte = pf_head;
        unsigned long long NL[NHALF], ORI[NHALF];
        SpanFlags fl = {0u, 0u, 0u, 0u};
        uint32_t dee_end = 0;                                            // wave-uniform: the first half ends with 'd'
        unsigned long long IDM[NHALF];                                   // wave-uniform: lanes whose span of this half holds the 'd' of a pair "d:"
#pragma unroll
        for (uint32_t h = 0; h < NHALF; ++h) {
            __builtin_amdgcn_s_setprio(P_A);
#pragma unroll
            for (uint32_t i = 0; i < NPF; ++i) *(uint4 *)(text + h * HALF + (i * WG + lane) * 16) = pf[i];
            if (h + 1 < NHALF) fetch_half(c0 + (h + 1) * HALF);           // the next half travels while this one is classified
            wave_sync();
            __builtin_amdgcn_s_setprio(0);
            fl.idf = 0u;
            classify_span(a, text, ndbm, tbm, h, h * WG + lane, c0, V, NL[h], ORI[h], fl);
            IDM[h] = ballot64(fl.idf != 0);
            if (h == 0) dee_end = rdlane(fl.dee_last, WG - 1);
            else if (dee_end != 0 && text[HALF] == ':') {                // (the pair straddles the halves: its 'd' is the first half's last byte)
                IDM[0] |= 1ull << 63;
                if (lane == WG - 1) { const uint32_t c = (fl.idsum >> 6) & 3u; fl.idsum = (fl.idsum & 0xFF00u) | ((c < 3u ? c + 1u : 3u) << 6) | 63u; }
            }
        }
        if (ballot64(fl.high != 0) != 0 && lane == 0) a.st->non_ascii = 1;
        const bool idf = (IDM[0] | IDM[1]) != 0;                         // some line of the stripe may hold an "id:f:" tag: the line phase finds which
        if (idf && lane == 0) {                                          // (kept in the line list's spare entries, not in scalar registers, until then)
            LINE[MAXL + 4] = (uint32_t)IDM[0]; LINE[MAXL + 5] = (uint32_t)(IDM[0] >> 32); LINE[MAXL + 6] = (uint32_t)IDM[1]; LINE[MAXL + 7] = (uint32_t)(IDM[1] >> 32);
        }
        tick(0);
        // does the stripe begin at a line start?  (wave-uniform)
        const uint32_t head = (head_byte == '\n') || (head_byte == '\r' && text[0] != '\n');
        uint32_t Vh = V;                                                 // bytes of the staged text this stripe looks at: all of them, or (dense text) the first half
        bool eof_h = at_eof;
        uint32_t lim2, n_s, n_own, tot_ori, l_first;
        bool last_stripe, long_line;
        unsigned long long next_pos;
        uint32_t RK[NHALF];                                              // ordinal of the lane's first terminator | of its first mark << 16, per half
        uint32_t head_own;
        for (uint32_t attempt = 0;; ++attempt) {
            // the last line start in the text looked at (a terminator at b starts a line at b + 1)
            uint32_t s_last = NONE32;
            {
                const unsigned long long b1 = ballot64(NL[1] != 0), b0 = ballot64(NL[0] != 0);
                if (b1 | b0) {
                    const uint32_t hh = b1 ? 1u : 0u;
                    const uint32_t L = 63u - (uint32_t)__builtin_clzll(b1 ? b1 : b0);
                    const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hh ? NL[1] : NL[0]), (int)L);
                    const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((hh ? NL[1] : NL[0]) >> 32), (int)L);
                    const unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
                    s_last = (hh * WG + L) * SPAN + (63u - (uint32_t)__builtin_clzll(m)) + 1u;
                } else if (head) s_last = 0;
            }
            // which lines this stripe handles, and where the next one begins
            const uint32_t own_lim = rend - c0 < (unsigned long long)Vh ? (uint32_t)(rend - c0) : Vh;
            lim2 = own_lim;
            last_stripe = false; long_line = false;
            next_pos = c0 + Vh;
            if (eof_h) last_stripe = true;                               // every line ends inside the staged text (the last one maybe without a terminator)
            else if (s_last == NONE32 || s_last < own_lo) { }            // no line starts here: the stripe lies inside one long line
            else if (c0 + s_last >= rend) last_stripe = true;            // the last line start is the next worker's: all of this worker's lines end in here
            else if (s_last > own_lo) { lim2 = s_last; next_pos = c0 + s_last; }   // the line at s_last has no end in here: the next stripe begins with it
            else long_line = true;                                       // a single line longer than the text looked at: exact path
            if (next_pos >= rend) last_stripe = true;

            // terminators and marks per lane and half -> their ordinals in the stripe: two wave prefix sums of two 16-bit fields each
            // (a half holds at most 4096 of either).  The line at the stripe's head is counted by lane 0.
            uint32_t in0 = ((uint32_t)__popcll(NL[0]) + (lane == 0 ? head : 0u)) | ((uint32_t)__popcll(ORI[0]) << 16);
            uint32_t in1 = (uint32_t)__popcll(NL[1]) | ((uint32_t)__popcll(ORI[1]) << 16);
            const uint32_t c0v = in0, c1v = in1;
            wave_incl_scan2(in0, in1);
            const uint32_t tot0 = rdlane(in0, WG - 1), tots = tot0 + rdlane(in1, WG - 1);
            RK[0] = in0 - c0v; RK[1] = tot0 + in1 - c1v;
            n_s = tots & 0xFFFFu; tot_ori = tots >> 16;                  // line starts and marks in the text looked at
            // line starts in front of position x (wave-uniform, x <= Vh): those this stripe handles lie in [own_lo, lim2)
            auto starts_before = [&](uint32_t x) -> uint32_t {
                if (x == 0) return 0u;
                const uint32_t q = x - 1u, slot = q >> 6, hh = slot >> 6, L = slot & 63u;   // terminators in front of q count
                const uint32_t rk = rdlane(hh ? RK[1] : RK[0], L) & 0xFFFFu;
                const uint32_t mlo = rdlane((uint32_t)(hh ? NL[1] : NL[0]), L), mhi = rdlane((uint32_t)((hh ? NL[1] : NL[0]) >> 32), L);
                const unsigned long long m = (((unsigned long long)mhi << 32) | mlo) & ((1ull << (q & 63u)) - 1ull);
                return rk + (uint32_t)__popcll(m) + (slot == 0 ? head : 0u);
            };
            l_first = starts_before(own_lo);
            n_own = starts_before(lim2) - l_first;
            head_own = head & (uint32_t)(own_lo == 0 && 0 < lim2);
            if (attempt || !(n_s > MAXL || tot_ori > CAP_O)) break;
            // The lists cannot hold the whole stripe (more than MAXL lines or CAP_O marks in 8 KB: short lines, paths of many nodes):
            // the stripe ends at the last line start up to which they can (the next stripe begins there), or, if there is none,
            // after its first half.  Both limits grow with the position, so the lane and bit to look for are the last that fit.
            uint32_t best[NHALF];
#pragma unroll
            for (uint32_t h = 0; h < NHALF; ++h) {
                const uint32_t sp = (h * WG + lane) * SPAN;
                const uint32_t sb = (RK[h] & 0xFFFFu) + ((h == 0 && lane == 0) ? head : 0u), ob = RK[h] >> 16;
                best[h] = 0;
                for (unsigned long long m = NL[h]; m; m &= m - 1) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(m);
                    const uint32_t ord = sb + (uint32_t)__popcll(NL[h] & ((1ull << b) - 1ull));
                    const uint32_t marks = ob + (uint32_t)__popcll(ORI[h] & ((2ull << b) - 1ull));
                    if (ord + 1u <= MAXL && marks <= CAP_O && sp + b + 1u <= Vh) best[h] = sp + b + 1u;
                }
            }
            uint32_t cut = 0;
            {
                const unsigned long long f1 = ballot64(best[1] != 0), f0 = ballot64(best[0] != 0);
                if (f1) cut = rdlane(best[1], 63u - (uint32_t)__builtin_clzll(f1));
                else if (f0) cut = rdlane(best[0], 63u - (uint32_t)__builtin_clzll(f0));
            }
            if (cut > own_lo && cut > 16u) Vh = cut;
            else if (Vh > HALF) Vh = HALF;
            else break;
            eof_h = false;
#pragma unroll
            for (uint32_t h = 0; h < NHALF; ++h) {                       // nothing at or behind Vh is looked at (the terminator at Vh - 1 stays)
                const unsigned long long keep = low_bits64(clamp64(Vh, (h * WG + lane) * SPAN));
                NL[h] &= keep; ORI[h] &= keep;
            }
        }
        // A line longer than the staged text (r04): if what runs past the stage is only the line's TAIL — its twelve columns and its path
        // lie in the staged 8 KB: the line phase finds out — the line stays here.  The tail is walked first, 4 KB per step from the
        // prefetch registers, no LDS, no byte classes: where is the terminator; does the tail hold a carriage return or the byte pair "d:"
        // (an id:f: tag: then the exact path decides, as for any tag the line phase cannot read) or a byte >= 0x80 (the host validates
        // UTF-8).  With a clean tail the stripe is handled like a text's last one (its line ends where the stage ends) and the next stripe
        // begins behind the terminator; otherwise, and beyond 256 KB, the line takes the exact path as before.
        // (Such a stripe is known below by long_line && eof_h: the two never hold together otherwise.)
        if (RARELY(long_line) && Vh == TEXT && !a.all_slow && tot_ori <= CAP_O && text[TEXT - 1] != 'd') {
            unsigned long long at = c0 + TEXT, E = ~0ull;
            bool clean = true;
            for (uint32_t it = 0; it < 64u && clean && E == ~0ull; ++it, at += HALF) {
                if (at >= a.n_bytes) { E = a.n_bytes; break; }           // the text ends inside the line (no terminator)
                fetch_half(at);
#pragma unroll
                for (uint32_t i = 0; i < NPF; ++i) {
                    if (E != ~0ull || !clean) break;
                    const uint4 v = pf[i];
                    const uint32_t nl = eq_mask16(v, 0x0A0A0A0Au), dee = eq_mask16(v, 0x64646464u), col = eq_mask16(v, 0x3A3A3A3Au);
                    const uint32_t bad = eq_mask16(v, 0x0D0D0D0Du) | (dee & ((col >> 1) | 0x8000u));   // (a 'd' at a piece's end: its ':' would be the next piece's)
                    if (ballot64(((v.x | v.y | v.z | v.w) & 0x80808080u) != 0) != 0 && lane == 0) a.st->non_ascii = 1;
                    uint32_t front = 0xFFFFu;                            // the piece's bytes in front of the terminator
                    const unsigned long long b = ballot64(nl != 0);
                    if (b != 0) {
                        const uint32_t L = (uint32_t)__builtin_ctzll(b), bit = (uint32_t)__builtin_ctz(rdlane(nl, L));
                        E = at + (unsigned long long)((i * WG + L) * 16u + bit);
                        front = lane < L ? 0xFFFFu : lane == L ? (1u << bit) - 1u : 0u;
                    }
                    if (ballot64((bad & front) != 0) != 0) clean = false;
                }
                if (E != ~0ull && E >= a.n_bytes) E = a.n_bytes;         // (cannot happen: the padding behind the text holds no terminator)
            }
            if (clean && E != ~0ull) {
                eof_h = true;
                next_pos = E < a.n_bytes ? E + 1ull : a.n_bytes;
                last_stripe = next_pos >= rend;
            }
        }
        if (!last_stripe) {
            if (RARELY(long_line && eof_h)) pf_head = (next_pos & 15ull) ? (uint32_t)'x' : (uint32_t)'\n';   // (the byte in front of the next stripe: a byte of the tail, or its terminator)
            else {
                const uint32_t d0 = (uint32_t)((next_pos & ~15ull) - c0);    // the next stripe begins inside (or right behind) this one's staged text
                pf_head = d0 ? (uint32_t)text[d0 - 1] : head_byte;
            }
            fetch_half(next_pos & ~15ull);
        }
        tick(1);

        if (a.all_slow || (long_line && !eof_h) || n_s > MAXL || tot_ori > CAP_O) {
            // the lists cannot hold this stripe / the caller wants the exact path: every owned line is deferred as it is
            if (n_own) {
                unsigned long long dbase = 0;
                if (lane == 0) { dbase = atomicAdd(&a.st->n_deferred, (unsigned long long)n_own); atomicAdd(&a.st->cause[DC_STRIPE], (unsigned long long)n_own); }
                dbase = __shfl(dbase, 0);
                if (lane == 0 && head_own) { if (dbase < a.deferred_cap) a.deferred[dbase] = c0; else atomicOr(&a.st->overflow, 1u); }   // (ordinal 0 = l_first)
#pragma unroll
                for (uint32_t h = 0; h < NHALF; ++h) {                   // the owned starts are the ordinals [l_first, l_first + n_own)
                    const uint32_t sp = (h * WG + lane) * SPAN;
                    uint32_t j = (RK[h] & 0xFFFFu) + ((h == 0 && lane == 0) ? head : 0u) - l_first;
                    for (unsigned long long m = NL[h]; m; m &= m - 1, ++j) {
                        if (j >= n_own) continue;                        // (unsigned: also the starts in front of l_first)
                        const unsigned long long d = dbase + j;
                        if (d < a.deferred_cap) a.deferred[d] = c0 + sp + (uint32_t)__builtin_ctzll(m) + 1u; else atomicOr(&a.st->overflow, 1u);
                    }
                }
            }
            wave_lines += n_own;
            wave_sync();
            if (last_stripe) break;
            pos = next_pos;
            continue;
        }

        if (DIAG(32u)) { wave_lines += n_own; if (last_stripe) break; pos = next_pos; continue; }   // measurement only: stop after B1
        __builtin_amdgcn_s_setprio(P_B2);
        // ---- B2: rank-indexed lists ----------------------------------------------------------------------
        if (lane == 0) {
            if (head) LINE[0] = 0u;
            LINE[n_s] = 0xFFFFu | (tot_ori << 16);
        }
#pragma unroll
        for (uint32_t h = 0; h < NHALF; ++h) {
            const uint32_t sp = (h * WG + lane) * SPAN;
            const uint32_t sb = (RK[h] & 0xFFFFu) + ((h == 0 && lane == 0) ? head : 0u);   // ordinal of the first line start this lane creates
            const uint32_t ob = RK[h] >> 16;
            uint32_t j = sb;
            for (unsigned long long m = NL[h]; m; m &= m - 1, ++j) {
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                LINE[j] = (sp + b + 1) | ((ob + (uint32_t)__popcll(ORI[h] & ((1ull << b) - 1ull))) << 16);
            }
            j = ob;
            // line that holds the mark; 0xFFFF: tail of a line of the previous stripe.  (32-bit halves were measured: more loops, more instructions)
            if (ballot64((NL[h] & (NL[h] - 1ull)) != 0) == 0) {            // wave-uniform: no span holds two terminators (lines of 64 bytes and more)
                const uint32_t t1 = NL[h] ? (uint32_t)__builtin_ctzll(NL[h]) : 64u;
                const uint32_t v0 = sp + ((sb - 1u) << 16);
                for (unsigned long long m = ORI[h]; m; m &= m - 1, ++j) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(m);
                    OPL_PUT(j, v0 + b + (b > t1 ? 0x10000u : 0u));
                }
            } else {
                for (unsigned long long m = ORI[h]; m; m &= m - 1, ++j) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(m);
                    OPL_PUT(j, (sp + b) | ((sb + (uint32_t)__popcll(NL[h] & ((1ull << b) - 1ull)) - 1u) << 16));
                }
            }
        }
        wave_sync();
        tick(2);

        __builtin_amdgcn_s_setprio(P_R1);
        // ---- rounds of up to LRW lines -----------------------------------------------------------------------------
        const uint32_t l_hi = l_first + n_own;
        for (uint32_t lbase = DIAG(1u) ? l_hi : l_first; lbase < l_hi; lbase += LRW) {   // wave-uniform trip count
            const uint32_t cnt = l_hi - lbase < LRW ? l_hi - lbase : LRW;
            const uint32_t obase = LINE[lbase] >> 16;
            // ---- R1: one line per lane --------------------------------------------------------------
            uint32_t status = ST_NONE, k = 0, s = 0, rel = 0, kall = 0, r_need_l = 0, r_need_r = 0, r_pend = 0;
            uint32_t id_span = NONE32, id_end = 0;         // (only when some span of the stripe holds a pair "d:")
            if (lane < cnt) {
                const uint32_t L = lbase + lane;
                const uint32_t l0 = LINE[L], l1 = LINE[L + 1];
                s = l0 & 0xFFFFu;
                const uint32_t nx = l1 & 0xFFFFu, o0 = l0 >> 16, o1 = l1 >> 16;
                kall = o1 - o0; rel = o0 - obase;
                status = ST_DEFER;
                if (nx != 0xFFFFu || eof_h) {                            // (always: the stripe handles only lines that end in it)
                    uint32_t e = nx != 0xFFFFu ? nx - 1 : Vh;
                    if (e > s && py_strip_space(text[e - 1])) {          // line.rstrip(): blanks, a CR, trailing tabs
                        do --e; while (e > s && py_strip_space(text[e - 1]));
                    }
                    // the twelve column ends, one after the other on the tab bitmap: a decimal column has its tab within ten bytes
                    // (one 32-bit window); read name, strand and path column may be any length
                    const uint32_t t0 = tab_from(tbm, s, e);
                    uint32_t wa = tab_window(tbm, t0 + 1);              // columns 2..4 (and, when they are short, the strand column's tab)
                    const uint32_t t1 = take_tab(wa, t0 + 1), t2 = take_tab(wa, t0 + 1), t3 = take_tab(wa, t0 + 1);
                    const uint32_t t4a = take_tab(wa, t0 + 1);
                    const uint32_t t4 = t4a < TEXT ? (t4a < e ? t4a : TEXT) : tab_from(tbm, t0 + 33 < TEXT ? t0 + 33 : TEXT, e);   // (nothing in the window: look further)
                    // The path column ends at the first tab behind the line's LAST orientation mark (a node name has at most 48 bytes).  Should a tab sit in front of that mark (marks in later columns), the piece of text
                    // between two marks that holds it is no node name, and the node pass sends the line to the exact path.
                    k = kall;
                    const bool kfit = k >= 1 && k <= KLONG;
                    const uint32_t m_first = OPL_POS(o0), m_last = OPL_POS(kfit ? o0 + k - 1 : o0);
                    uint32_t t5 = tab_near(tbm, m_last + 1);
                    if (RARELY(t5 == TEXT)) {                             // (a name of 32..64 bytes: the tab is at most 65 bytes behind the mark)
                        t5 = tab_near(tbm, m_last + 33 < TEXT ? m_last + 33 : TEXT);
                        if (t5 == TEXT) { const uint32_t f2 = tab_near(tbm, m_last + 65 < TEXT ? m_last + 65 : TEXT); t5 = f2 == m_last + 65u ? f2 : TEXT; }
                    }
                    uint32_t wb = tab_window(tbm, t5 + 1);              // columns 7..9
                    const uint32_t t6 = take_tab(wb, t5 + 1), t7 = take_tab(wb, t5 + 1), t8 = take_tab(wb, t5 + 1);
                    uint32_t wc = tab_window(tbm, t8 + 1);              // columns 10..12
                    const uint32_t t9 = take_tab(wc, t8 + 1), t10 = take_tab(wc, t8 + 1), t11n = take_tab(wc, t8 + 1);
                    const uint32_t t11 = t11n < e ? t11n : e;
                    bool ok = t10 < e;                                   // eleven tabs inside the (stripped) line: twelve columns
                    if (RARELY(long_line)) ok = ok && t11n < e;          // (r05; a line that runs past the staged text: its twelfth column must END in the stage — "60" staged, "x" in the tail is no number)
                    // the nine decimal columns: 1..9 characters each, nothing but digits in them
                    ok &= (t1 - t0 - 2u <= 8u) & (t2 - t1 - 2u <= 8u) & (t3 - t2 - 2u <= 8u);
                    ok &= (t6 - t5 - 2u <= 8u) & (t7 - t6 - 2u <= 8u) & (t8 - t7 - 2u <= 8u);
                    ok &= (t9 - t8 - 2u <= 8u) & (t10 - t9 - 2u <= 8u) & (t11 - t10 - 2u <= 8u);
                    // From here on straight code on positions made harmless when the line is not `ok` (every LDS read goes out at
                    // once instead of one dependent round trip per nested test).
                    const uint32_t u0 = ok ? t0 : 0u, u3 = ok ? t3 : 1u, u4 = ok ? t4 : 0u, u5 = ok ? t5 : 0u, u6 = ok ? t6 : 2u, u7 = ok ? t7 : 4u;
                    const uint32_t u8 = ok ? t8 : 6u, u9 = ok ? t9 : 8u, u10 = ok ? t10 : 10u, u11 = ok ? t11 : 12u;
#ifdef SVJG_W16
                    const bool digits = true;                            // (measurement variant: not tested)
#else
                    const bool digits = bits_clear(ndbm, u0 + 1, u3 - u0 - 1) & bits_clear(ndbm, u5 + 1, u11 - u5 - 1);
#endif
                    const bool alen0 = field_is_zero(text, u9 + 1, u10 - u9 - 1);   // Alen == 0: ZeroDivisionError (no id:f: tag in this stripe): exact path decides
                    // path column (t4, t5): the first orientation mark of the line right after t4, the last one in front of t5
                    // check_bkpt_overlap (filter-alignments.py:258-273) for a link of this line reads
                    //   sum(len up to the left node) - Ts >= d_over  and  sum(len from the right node) - (Tlen - Te - 1) >= d_over:
                    // the two right-hand sides are fixed per line
                    // (columns of at most nine digits and d_over < 2^31, svjg_load_graph: everything fits 32 bits)
                    const uint32_t tlen = field_val(text, u5 + 1, u6 - u5 - 1), ts = field_val(text, u6 + 1, u7 - u6 - 1), te = field_val(text, u7 + 1, u8 - u7 - 1);
                    const bool ok_cols = ok && digits && kfit && u5 > u4 + 1 && m_first == u4 + 1 && m_last < u5;
                    ok = ok_cols && !alen0;
                    if (RARELY(idf)) {                                   // which spans of this line hold a pair "d:" (decided behind the block, all lanes together)
                        const uint32_t lend = nx != 0xFFFFu ? nx : Vh;  // spans [s >> 6, (next line start - 1) >> 6] of the 128 of the stripe
                        const uint32_t sa = s >> 6, sb = (lend - 1u) >> 6;
                        const unsigned long long m0 = low_bits64(sb + 1u < 64u ? sb + 1u : 64u) & ~low_bits64(sa < 64u ? sa : 64u);
                        const unsigned long long m1 = low_bits64(sb >= 64u ? sb - 63u : 0u) & ~low_bits64(sa > 64u ? sa - 64u : 0u);
                        unsigned long long f0 = ((unsigned long long)LINE[MAXL + 4] | ((unsigned long long)LINE[MAXL + 5] << 32)) & m0;
                        unsigned long long f1 = ((unsigned long long)LINE[MAXL + 6] | ((unsigned long long)LINE[MAXL + 7] << 32)) & m1;
                        auto take = [&](uint32_t sp) -> uint32_t {       // is span sp flagged?  (and no longer, afterwards)
                            unsigned long long &f = sp < 64u ? f0 : f1;
                            const unsigned long long bit = 1ull << (sp & 63u);
                            const uint32_t r = (f & bit) ? 1u : 0u;
                            f &= ~bit;
                            return r;
                        };
                        const uint32_t fa = take(sa), fb = take(sb);     // the line's first and last span may hold a neighbour's pair
                        const uint32_t n_int = (uint32_t)__popcll(f0) + (uint32_t)__popcll(f1);   // the spans in between hold only this line's
                        const uint32_t isp = f0 ? (uint32_t)__builtin_ctzll(f0) : f1 ? 64u + (uint32_t)__builtin_ctzll(f1) : 0u;
                        if (fa | fb | n_int) id_span = fa | (fb << 1) | ((n_int < 3u ? n_int : 3u) << 2) | (isp << 4) | (ok_cols ? 1u << 11 : 0u);
                        id_end = e | (lend << 16);
                    }
                    if (!ok && status == ST_DEFER && kall > KLONG) status = ST_DEFER + DC_LONG_PATH;
                    else if (!ok && status == ST_DEFER && t5 == TEXT && t4 < TEXT && kfit) status = ST_DEFER + DC_NAME;   // (no tab within 65 bytes of the last mark: a name beyond 64 bytes)
                    if (ok_cols) {
                        if (ok) status = k >= 2 ? ST_OK : ST_NOHIT;
                        r_need_l = ts + g.d_over;
                        r_need_r = tlen + g.d_over > te + 1u ? tlen + g.d_over - te - 1u : 0u;
                        r_pend = t5;
                    }
                }
            }
            if (RARELY(idf)) {
                // "id:f:" anywhere in a line changes what the reference does with it (:193-196): float() of the text behind the LAST "id:f:"
                // up to the next tab may raise, and Alen == 0 no longer does.  The byte classes know every pair "d:" — which every such tag
                // holds —: IDM = the 64-byte spans with the 'd' of a pair, idsum = how many pairs a span has and where its last one is.  A line
                // whose spans hold exactly one pair is decided here: the pair in another line of the span, or no "id:f:" around it -> an ordinary
                // line; a tag whose value is digits with at most one '.' in or next to them (what aligners write) -> an ordinary line whose
                // Alen may be zero; any other value, and a line with several pairs, takes the exact path.
                const bool has = id_span != NONE32;
                const uint32_t lend = id_end >> 16, e = id_end & 0xFFFFu;
                const uint32_t sa = has ? s >> 6 : 0u, sb = has ? (lend - 1u) >> 6 : 0u, isp = has ? (id_span >> 4) & 127u : 0u;
                auto summary = [&](uint32_t sp) -> uint32_t { return ((uint32_t)__shfl((int)fl.idsum, (int)(sp & 63u)) >> (sp & 64u ? 8 : 0)) & 0xFFu; };
                const uint32_t sum_a = summary(sa), sum_b = summary(sb), sum_i = summary(isp);   // (all lanes: the values come from the lanes that own the spans)
                if (has) {
                    uint32_t mine = 0, q = 0;                            // pairs whose 'd' lies in this line: how many (if that can be told), where the only one is
                    bool unsure = false;
                    if (id_span & 1u) {                                  // first span: the pairs in front of the line's start are a neighbour's
                        const uint32_t qa = sa * SPAN + (sum_a & 63u);
                        if (qa >= s && qa < lend) { if ((sum_a >> 6) == 1u) { ++mine; q = qa; } else unsure = true; }
                        else if (qa >= lend && (sum_a >> 6) != 1u) unsure = true;   // (a line inside one span, several pairs in it)
                    }
                    if (id_span & 2u) {                                  // last span (not the first): a single pair behind the line's end is a neighbour's
                        const uint32_t qb = sb * SPAN + (sum_b & 63u);
                        if ((sum_b >> 6) != 1u) unsure = true;
                        else if (qb < lend) { ++mine; q = qb; }
                    }
                    const uint32_t n_int = (id_span >> 2) & 3u;
                    if (n_int == 1u) { if ((sum_i >> 6) == 1u) { ++mine; q = isp * SPAN + (sum_i & 63u); } else unsure = true; }
                    else if (n_int) unsure = true;
                    uint32_t verdict = 2u;                               // 0: no tag in this line, 1: a tag with a plain value, 2: the exact path decides
                    if (!unsure && mine == 0u) verdict = 0u;
                    else if (!unsure && mine == 1u) {
                        // (r05) a line that runs past the staged text (long_line: its tail is walked apart, above) may hold a tag that the stage's end
                        // cuts — its "id:f:" or its value continue in the tail, which only rules out a pair "d:" of its own —: the exact path decides
                        if (!(q > s && q + 4u <= e && text[q - 1u] == 'i' && text[q + 2u] == 'f' && text[q + 3u] == ':')) verdict = (RARELY(long_line) && q + 4u > e) ? 2u : 0u;   // (inside the stripped line, like `in line`)
                        else {
                            const uint32_t v0 = q + 4u;                  // value = text[v0, first tab or end of the stripped line)
#ifdef SVJG_W16
                            const uint32_t tw = 0u, nw = 0u;             // (measurement variant: every tag takes the exact path)
#else
                            const uint32_t tw = tab_window(tbm, v0), nw = tab_window(ndbm, v0);
#endif
                            uint32_t n = tw ? (uint32_t)__builtin_ctz(tw) : 32u;
                            if (e - v0 < n) n = e - v0;
#ifdef SVJG_W16
                            n = 0u;
#endif
                            if (RARELY(long_line) && v0 + n >= e) n = 0u;    // (no tab in front of the stage's end: the value goes on in the tail)
                            if (n >= 1u && n <= 31u) {
                                const uint32_t nd = nw & ((1u << n) - 1u);
                                if (nd == 0u) verdict = 1u;
                                else if ((nd & (nd - 1u)) == 0u && n >= 2u && text[v0 + (uint32_t)__builtin_ctz(nd)] == '.') verdict = 1u;
                            }
                        }
                    }
                    if (verdict == 2u) { if (status < ST_DEFER || status == ST_DEFER + DC_COLUMNS) status = ST_DEFER + DC_IDF; }
                    else if (verdict == 1u && (id_span & (1u << 11)) && status == ST_DEFER + DC_COLUMNS) status = k >= 2 ? ST_OK : ST_NOHIT;   // (it was only Alen == 0 that held the line back)
                }
            }
            if (status != ST_OK) k = 0;
            RL[lane] = make_uint4(r_need_l, r_need_r, rel | (k << 16) | (status << 24), r_pend);
            wave_sync();
            tick(3);
            if (DIAG(2u)) continue;                                      // measurement only: stop after R1
            // ---- node passes: up to 64 consecutive marks that cover whole lines; one mark (path node) per lane ----------------
            const wmask ok_lines = m_eq(status, ST_OK);                  // (lanes beyond cnt: ST_NONE)
            const uint32_t relend = rel + kall;
            // A line of more than 64 nodes (wave-uniform state, all of it in three scalars; the line is lane i0's).  lsub = 0: none in hand; else
            // bit 0 set, L_SWEEP1 the line's second sweep, L_MEASURE its first sweep only measures any more, L_FINAL the line's last
            // sub-pass, bits 4-6 lD0 (which way its ids run: 1 / 2 rise / fall, 3 they turn, 4 a name comes twice), bits 8-15 lP (index of
            // the sub-pass's first node), bits 16-23 lR (the first link not counted yet when the line went over to two sweeps), L_ONE this
            // sub-pass of the first sweep decides links as it goes; lS = length of the path in front of the sub-pass's first node; lTOT =
            // the path's total length (second sweep).
            constexpr uint32_t L_SWEEP1 = 2u, L_MEASURE = 4u, L_FINAL = 8u, L_ONE = 1u << 24;
            uint32_t lsub = 0, lS = 0, lTOT = 0;
#define lD0 ((lsub >> 4) & 7u)
#define lP ((lsub >> 8) & 0xFFu)
#define lR ((lsub >> 16) & 0xFFu)
#define SET_D0(v) (lsub = (lsub & ~0x70u) | ((uint32_t)(v) << 4))
#define SET_P(v) (lsub = (lsub & ~0xFF00u) | ((uint32_t)(v) << 8))
#define SET_R(v) (lsub = (lsub & ~0xFF0000u) | ((uint32_t)(v) << 16))
            for (uint32_t i0 = 0;;) {
                uint32_t p0, n_pass;
                wmask okl;
                if (!RARELY(lsub)) {
                    if (i0 >= cnt) break;
                    p0 = rdlane(rel, i0);
                    const wmask nofit = m_gt(relend - p0, 64u) & low_bits64(cnt) & ~low_bits64(i0);
                    const uint32_t i1 = nofit ? (uint32_t)__builtin_ctzll(nofit) : cnt;
                    if (RARELY(i1 == i0)) {                              // a line with more than 64 marks: sub-passes, if its columns were fine
                        if ((ok_lines >> i0) & 1ull) { lsub = 1u; lS = 0; } else ++i0;
                        continue;
                    }
                    n_pass = rdlane(relend, i1 - 1) - p0;
                    okl = ok_lines & low_bits64(i1) & ~low_bits64(i0);
                    i0 = i1;
                    if (!okl) continue;                                  // no line of the pass has a path to look at
                } else {
                    const uint32_t K = rdlane(kall, i0);
                    p0 = rdlane(rel, i0) + lP;
                    n_pass = K - lP < 64u ? K - lP : 64u;
                    lsub = (lsub & ~(L_FINAL | L_ONE)) | (K - lP <= 64u ? L_FINAL : 0u);
                    okl = 1ull << i0;
                }
                __builtin_amdgcn_s_setprio(P_LOAD);
                tally(10, 1); tally(5, n_pass); if (lsub) { tally(11, 1); tally(15, n_pass); if (lsub & L_SWEEP1) tally(14, 1); }
                // -- the node of this lane: line, index in the line, name.  Every lane runs the same straight code on indices that
                //    are safe to read (a lane beyond the pass looks at the pass's first mark); `live` says whose results count --
                const wmask act_m = low_bits64(n_pass);
                const bool act = in_mask(act_m);
                const uint32_t o = obase + p0 + (act ? lane : 0u);
#ifdef SVJG_W16
                uint2 op2;
                { const uint32_t pp = *(const u32_any *)(OPLP + o); op2.x = (pp & 0xFFFFu) | (((uint32_t)OPLL[o] - 1u) << 16); op2.y = pp >> 16; }
#else
                const uint2 op2 = *(const uint2 *)(OPL + o);             // (8-byte aligned or not: two dwords)
#endif
                const uint32_t opv = op2.x & 0xFFFFu;
                const uint32_t ln = ((op2.x >> 16) - lbase) & (LRW - 1u);
                const uint4 rl = RL[ln];
                const uint32_t meta = rl.z, need_l = rl.x, need_r = rl.y;
                wmask live_m = act_m & m_eq(meta >> 24, ST_OK);
                bool live = in_mask(live_m);
                uint32_t lnb = live ? (meta & 0xFFFFu) - p0 : 0u, lk = live ? (meta >> 16) & 0xFFu : 0u, j = live ? lane - lnb : 0u;
                const uint32_t na0 = opv + 1u;
                const uint32_t len = ((j + 1 < lk) ? (op2.y & 0xFFFFu) : rl.w) - na0;
                const uint32_t oribit = text[opv] == '<' ? 1u : 0u;
                const wmask probe_m = live_m & m_le(len - 1u, 4u * NAME_WORDS - 1u);   // names of 1..48 bytes (49..64: the cold block below; longer ones: exact path)
                const bool probe = in_mask(probe_m);
                uint32_t d[8];
                tick_mem(8);                                             // (list and per-line record read)
                const uint32_t plen = probe ? len : 8u;                  // (keeps the longer names' window addresses inside the staged text)
                uint64_t h = name_words_head(text, na0, len, (probe_m & m_lt(len, 8u)) != 0, d);   // the first three windows of the name
                if (RARELY(probe_m & m_gt(len, 24u))) h += name_words_tail(text, na0, plen, d);      // (wave-uniform: node names of the usual length fit three windows)
                const bool long_names = (probe_m & m_gt(len, 32u)) != 0;                     // (wave-uniform: some name of the pass has 33..48 bytes)
                if (RARELY(long_names)) h += name_words_tail2(text, na0, plen);
                // -- perfect hash of the names: the bucket's displacement (a small, cache-resident array), then the ONE record
                //    the name can be in: 64 bytes with the spelling, id, length and the node's commonest links --
                uint32_t dsp = 0;
                if (probe) dsp = g.name_disp[name_bucket(h, g.name_buckets)];
                tick_mem(9);                                             // (name bytes, hash, displacement)
                // (lanes without a name load nothing: their record registers hold whatever was there and are looked at under `probe` only —
                //  a later `go` implies a matched node)
                uint4 r0, r1, r2, r3;
                asm volatile("" : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r1.w));
                asm volatile("" : "=v"(r2.x), "=v"(r2.y), "=v"(r2.z), "=v"(r2.w), "=v"(r3.x), "=v"(r3.y), "=v"(r3.z), "=v"(r3.w));
                if (probe) {
                    const uint4 *e = (const uint4 *)(g.name_tab + (size_t)name_slot(h, dsp, g.name_slots) * 16);
                    r0 = e[0]; r1 = e[1]; r2 = e[2]; r3 = e[3];
                }
                __builtin_amdgcn_s_setprio(P_REST);
                uint32_t id = NONE32, lbp = 0;
                uint32_t row_inline = 0;
                // id << 8 | flags << 6 | byte length - 1, length in bp; hazard-prone name / unknown alt length: exact path
                bool same = probe && name_match(r0, r1, r2, d, len);
                if (RARELY(long_names)) { uint32_t f[4]; name_words_far(text, na0, plen, f); same = same && (len <= 32u || ((r2.z ^ f[0]) | (r2.w ^ f[1]) | (len > 40u ? (r3.x ^ f[2]) | (r3.y ^ f[3]) : 0u)) == 0u); }
                if (same && r1.z != 0xFFFFFFFFu && !(r1.z & (NAME_FLAG_HAZARD | NAME_FLAG_NOLEN))) { id = r1.z >> NAME_ID_SHIFT; lbp = r1.w & 0x7FFFFFFFu; row_inline = r1.w >> 31; }
                // an unknown node, or one so long that 64 of them could overflow the 32-bit path sums: the line takes the exact
                // path.  The lanes that see it say so in the line's record, and every lane of the pass reads its line's record again
                // (all nodes of a line sit in this pass — or the line is a long one, none of whose links has been counted yet: what its
                // earlier sub-passes found waits in the worker's log).  Ordinary text never gets here.
                {
                    // One test on the path of ordinary text — is any node of the pass unknown, or 2^25 bp long and more — in front of three
                    // things ordinary text never needs (in this order):
                    //  * r06 — names of 49..64 bytes (contigs named like assemblies name their scaffolds) are looked up HERE by the windows of their
                    //    LAST 48 bytes; their first len - 48 bytes enter the hash and are held against a table of their own (g.name_pfx, four
                    //    words per node id; svjg_line.h: name_prefix_words).  A second, dependent trip to the tables for such a pass, nothing
                    //    for every other (woven into the lookup above it cost the headline 0.9 %: profiles/r06/experiments/names_49_to_64_bytes.txt).
                    //  * r06 — a node of 2^25 bp and more (GRCh37 has SV-free stretches of that order: a whole-genome graph's node Y:25 Mbp-59 Mbp,
                    //    every contig without an SV) no longer sends its line away.  The wave's prefix sum below wraps modulo 2^32 across
                    //    LINES, which the per-line differences undo; what must hold is that each line's OWN path stays below 2^32 bp.  The
                    //    same prefix sum over the lengths >> 6 (at most 2^25 each: no wrap over 64 lanes) bounds every line's total from
                    //    above: (sum of len >> 6) + nodes < 2^26  =>  path < 2^32.  A sub-pass of a line of > 64 nodes keeps the old rule
                    //    (its running total is checked for wrapping ONCE per sub-pass).
                    //  * an unknown node, or a path that long: the line takes the exact path.  The lanes that see it say so in the line's
                    //    record, and every lane of the pass reads its line's record again (all nodes of a line sit in this pass — or the
                    //    line is a long one, none of whose links has been counted yet: what its earlier sub-passes found waits in the
                    //    worker's log).
                    wmask bad = live_m & m_eq(id, NONE32);
                    wmask big = live_m & m_ge(lbp, 1u << 25);
                    if (RARELY(bad | big)) {
#ifndef SVJG_NO_VLONG
                        {
                            const wmask vl_m = bad & m_le(len - (4u * NAME_WORDS + 1u), NAME_MAX_BYTES - 4u * NAME_WORDS - 1u);
                            if (vl_m != 0ull && g.name_pfx != nullptr) {
                                const bool vl = in_mask(vl_m);
                                const uint32_t a1 = vl ? na0 + len - 4u * NAME_WORDS : na0;     // (the others: an address inside the staged text)
                                // The twelve window words of the last 48 bytes are those bytes' words in the order 0..5, 10, 11, 6..9 (svjg_line.h:
                                // name_windows with len = 48), so hash and compare walk the staged text word by word — one temporary each, the
                                // record goes into the pass's own r0..r3 (a lane without a record holds nothing there): no scratch memory.
                                const u32_any *sw = (const u32_any *)(text + a1);
                                const u32_any *pwd = (const u32_any *)(text + na0);
                                const int32_t npre = vl ? (int32_t)(len - 4u * NAME_WORDS) : 1;       // bytes in front of the last 48
                                auto pword = [&](int32_t i) -> uint32_t { const int32_t k = npre - 4 * i; return k >= 4 ? pwd[i] : k <= 0 ? 0u : (pwd[i] & ((1u << (8 * k)) - 1u)); };
                                uint64_t hh = (uint64_t)len * 0x7FEB352Du;
                                hh += (uint64_t)sw[0] * 0x9E3779B1u; hh += (uint64_t)sw[1] * 0x85EBCA77u; hh += (uint64_t)sw[2] * 0xC2B2AE3Du; hh += (uint64_t)sw[3] * 0x27D4EB2Fu;
                                hh += (uint64_t)sw[4] * 0x165667B1u; hh += (uint64_t)sw[5] * 0xD3A2646Du; hh += (uint64_t)sw[10] * 0xFD7046C5u; hh += (uint64_t)sw[11] * 0xB55A4F09u;
                                hh += (uint64_t)sw[6] * 0x94D049BBu; hh += (uint64_t)sw[7] * 0xBF58476Du; hh += (uint64_t)sw[8] * 0x2545F491u; hh += (uint64_t)sw[9] * 0x9FB21C65u;
                                hh += (uint64_t)pword(0) * 0xA24BAED5u; hh += (uint64_t)pword(1) * 0x9FB21C65u; hh += (uint64_t)pword(2) * 0xE7037ED1u; hh += (uint64_t)pword(3) * 0x8EBC6AF1u;
                                if (vl) {
                                    const uint4 *e = (const uint4 *)(g.name_tab + (size_t)name_slot(hh, g.name_disp[name_bucket(hh, g.name_buckets)], g.name_slots) * 16);
                                    r0 = e[0]; r1 = e[1]; r2 = e[2]; r3 = e[3];
                                }
                                uint32_t diff = vl ? ((r1.z & NAME_LEN_MASK) ^ (len - 1u)) : 1u;
                                diff |= (r0.x ^ sw[0]) | (r0.y ^ sw[1]) | (r0.z ^ sw[2]) | (r0.w ^ sw[3]) | (r1.x ^ sw[4]) | (r1.y ^ sw[5]);
                                diff |= (r2.x ^ sw[10]) | (r2.y ^ sw[11]) | (r2.z ^ sw[6]) | (r2.w ^ sw[7]) | (r3.x ^ sw[8]) | (r3.y ^ sw[9]);
                                if (diff == 0u && r1.z != 0xFFFFFFFFu) {
                                    const uint4 pq = *(const uint4 *)(g.name_pfx + (size_t)(r1.z >> NAME_ID_SHIFT) * NAME_PFX_WORDS);
                                    diff = (pq.x ^ pword(0)) | (pq.y ^ pword(1)) | (pq.z ^ pword(2)) | (pq.w ^ pword(3));
                                    if (diff == 0u && !(r1.z & (NAME_FLAG_HAZARD | NAME_FLAG_NOLEN))) {
                                        id = r1.z >> NAME_ID_SHIFT; lbp = r1.w & 0x7FFFFFFFu; row_inline = r1.w >> 31;
                                        h = hh;                              // (the hash the link table is asked with; r3.zw is the name's one inline link)
                                    }
                                }
                                bad = live_m & m_eq(id, NONE32);
                                big = live_m & m_ge(lbp, 1u << 25);
                            }
                        }
#endif
#ifdef SVJG_NO_BIG                                                         /* measurement variant: the r05 rule */
                        bad |= big;
#else
                        if (big) {
                            if (RARELY(lsub)) bad |= big;
                            else {
                                const uint32_t l6 = live ? (lbp >> 6) + 1u : 0u, s6 = wave_incl_scan(l6);
                                const uint32_t first6 = (uint32_t)__shfl((int)(s6 - l6), (int)lnb);
                                const uint32_t last6 = (uint32_t)__shfl((int)s6, (int)(lane + (lk ? lk - 1u - j : 0u)));
                                bad |= live_m & m_ge(last6 - first6, 1u << 26);
                            }
                        }
#endif
                        if (bad) {
                            if (in_mask(bad)) ((uint32_t *)&RL[ln])[2] = (meta & 0x00FFFFFFu) | ((ST_DEFER + DC_NAME) << 24);
                            wave_sync();
                            live_m &= m_eq(((const uint32_t *)&RL[ln])[2] >> 24, ST_OK);
                            live = in_mask(live_m);
                        }
                    }
                }
                if (!live) { j = 0; lk = 0; lnb = 0; id = NONE32; lbp = 0; }
                tick(4);
                // -- running path length of the line (inclusive) = wave prefix sum minus what precedes the line's first node --
                const uint32_t gsum = wave_incl_scan(lbp);               // (wraps modulo 2^32 across lines; every line's own total is below 2^32: above)
                const uint32_t gfirst = (uint32_t)__shfl((int)(gsum - lbp), (int)lnb);
                const uint32_t glast = (uint32_t)__shfl((int)gsum, (int)(lane + (lk ? lk - 1 - j : 0u)));
                uint32_t pre = gsum - gfirst, tot = glast - gfirst;
                // -- first occurrence of every name in its line (the reference's list.index / str.split quirks): every lane looks at
                //    the lanes below it, one DPP wave shift per distance (no LDS round trips), as far as the longest line of the pass
                //    reaches.  key = id | line << 26: lanes of other lines never compare equal --
                // A line whose node ids rise all the way, or fall all the way, cannot come back to a node (the ids follow the genome): only
                // when some line of the pass does neither do the lanes of THOSE lines search, as far as the longest of them reaches.
                const wmask step_m = m_lt(j + 1u, lk) & low_bits64(n_pass - 1u);   // lanes with a step to the next node of their line inside the pass (a dead lane: lk = 0)
                const uint32_t nxv = lane_above((id << 1) | oribit);
                const uint32_t idl = id, idr = nxv >> 1;
                uint32_t f = lane;
                const uint32_t dir = idr > id ? 1u : idr < id ? 2u : 0u;
                bool l_fail = false;
                if (RARELY(lsub)) {
                    // -- a sub-pass of a long line: the path's length in front of it comes from the sub-passes before; the first sweep counts
                    //    as it goes (its hits wait in the log), and asks ONCE, at the line's last node, whether a name has come twice (below) --
                    pre = gsum + lS; tot = lTOT;
                    l_fail = lS + rdlane(gsum, n_pass - 1u) < lS;        // (a path of 4 Gbp and more: the sums here are 32 bits wide)
                    if (!(lsub & L_SWEEP1)) {
                        // every node's id and orientation (where the tab bitmap was: the line phase is over) and the path length behind it (the
                        // worker's words of global memory): what the question at the line's end, and a second sweep, look at
                        if (live) {
                            tbm[lP + lane] = id | (oribit << 31);
                            __hip_atomic_store(a.long_pre + (size_t)blockIdx.x * LONG_WORDS + lP + lane, pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                {
                    const uint32_t dprev = lane_below(dir);
                    const wmask oddm = step_m & (m_eq(dir, 0u) | (m_ge(j, 1u) & m_ne(dir, dprev)));
                    if (RARELY(oddm) && !RARELY(lsub) && !DIAG(256u)) {      // (a long line: r06, see long_line_may_repeat; DIAG 256, measurement only: no search at all)
                        // the lanes of the lines that have such a step
                        const wmask search_m = live_m & ballot64(((low_bits64(lk) << lnb) & oddm) != 0ull);
                        const bool search = in_mask(search_m);
                        const uint32_t key = search ? (id | (ln << 26)) : NONE32;
                        const uint32_t jl = j < lane ? j : lane;             // nodes of the line below this lane, in this pass
                        uint32_t y = key;
                        for (uint32_t dd = 1; search_m & m_ge(jl, dd); dd += 4) {
                            y = lane_below_or(y, NONE32); if (y == key) f = lane - dd;
                            y = lane_below_or(y, NONE32); if (y == key) f = lane - dd - 1u;
                            y = lane_below_or(y, NONE32); if (y == key) f = lane - dd - 2u;
                            y = lane_below_or(y, NONE32); if (y == key) f = lane - dd - 3u;
                        }
                        if (!search) f = lane;                               // (a lane that did not look: NONE32 == NONE32 means nothing)
                    }
                }
                const bool revisits = m_ne(f, lane) != 0;                // wave-uniform: some line of the pass comes back to a node
                if (RARELY(lsub) && !(lsub & L_SWEEP1)) {
                    // the first sweep over a long line.  A name the table does not hold (its record says so already) -> the exact path.
                    // While the ids run one way and no name has come twice, the links are counted as the sweep goes (one_sweep): the path's
                    // total length is not known yet, but the part measured so far is a lower bound of it, and a link whose right-hand overlap
                    // holds against the bound holds; the next sub-pass begins at the first link that cannot be decided yet.  Once the ids turn,
                    // or such a link is the sub-pass's first, the rest of the sweep only measures (0x200) and a second sweep counts what the
                    // first has not (from link lR on).  Whatever the first sweep finds is held back (the worker's log) until the line's last
                    // name is known: the line may still turn out to be the exact path's, as a whole.
                    if (live_m == 0ull || l_fail) {
                        if (live_m != 0ull) { if (lane == 0) ((uint32_t *)&RL[i0])[2] = (rdlane(meta, 0) & 0x00FFFFFFu) | ((ST_DEFER + DC_LONG_PATH) << 24); wave_sync(); }
                        ++i0; lsub = 0;
                        continue;
                    }
                    // r06 — does a name come twice?  Asked ONCE, when the line's last node is known, of a bitmap of the ids (long_line_may_repeat)
                    // instead of every node of every sub-pass searching all nodes in front of it (0.32 of the long-read block's 1.89 ms:
                    // profiles/r06/experiments/long_read_block.txt).  Until then the sweep takes every name for a first occurrence; its hits
                    // wait in the log anyway.  If one may come twice, nothing of that stays (lR = 0) and the second sweep counts every link with
                    // every first occurrence looked up (lD0 = 4).
                    if (lsub & L_FINAL) {
                        wave_sync();                                     // (the ids of this sub-pass are in LDS)
                        if (long_line_may_repeat(tbm, rdlane(kall, i0), lane)) { SET_D0(4u); lsub |= L_MEASURE; SET_R(0u); }
                    }
                    if (!(lsub & L_MEASURE)) {
                        tally(12, 1);
                        lsub |= L_ONE;
                        tot = lS + rdlane(gsum, n_pass - 1u);
                    } else {
                        tally(13, 1);
                        if (lsub & L_FINAL) { lTOT = lS + rdlane(gsum, n_pass - 1u); lS = 0; lsub = (lsub & 0xFF0070u) | 1u | L_SWEEP1; }   // (the second sweep begins at the line's first node; lR and lD0 stay; what the first sweep has found — links [0, lR) — waits in the log until the second is through)
                        else { lS += rdlane(gsum, n_pass - 2u); SET_P(lP + n_pass - 1u); }
                        continue;
                    }
                }
                // -- the link this node -> next node: the reference evaluates name and strand of the FIRST occurrence of both
                //    (str.split / list.index, filter-alignments.py:206, :269-271); equal names have equal ids and hashes --
                uint32_t fl = lane, fr = lane + 1u, pre_l = pre, pre_rx = pre, orl = oribit, orr = nxv & 1u;
                wmask moved_m = 0;
                if (RARELY(revisits)) {
                    fl = f; fr = lane_above(f);
                    moved_m = m_ne(fl, lane) | m_ne(fr, lane + 1u);
                    pre_l = (uint32_t)__shfl((int)pre, (int)fl);
                    pre_rx = (uint32_t)__shfl((int)pre, (int)((fr - 1u) & 63u));
                    orl = (uint32_t)__shfl((int)oribit, (int)fl); orr = (uint32_t)__shfl((int)oribit, (int)(fr & 63u));
                }
                if (RARELY(lsub) && lD0 == 4u) {
                    // sweep 1 of a long line that comes back to a node: every node's first occurrence over the WHOLE line (list.index), its
                    // orientation there and the path length in front of it — the ids of all nodes wait in LDS, the running lengths in the
                    // worker's words of global memory (written in sweep 0 by this wave; read past the L1)
                    const uint32_t t63 = lP, K = rdlane(kall, i0), upto = K < t63 + 64u ? K : t63 + 64u;
                    const uint32_t *IDS = tbm;
                    const uint32_t jg = t63 + lane;
                    uint32_t fg = jg;
                    for (uint32_t m = 0; m < upto; ++m) { const uint32_t w = IDS[m] & 0x00FFFFFFu; if (w == id && m < fg) fg = m; }
                    const uint32_t fgr = lane_above(fg);
                    moved_m = step_m & (m_ne(fg, jg) | m_ne(fgr, jg + 1u));
                    pre_l = pre; pre_rx = pre; orl = oribit; orr = nxv & 1u; fr = lane + 1u;
                    if (moved_m) long_words_sync();
                    if (in_mask(moved_m)) {
                        const uint32_t *P = a.long_pre + (size_t)blockIdx.x * LONG_WORDS;
                        pre_l = __hip_atomic_load(P + fg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        pre_rx = fgr ? __hip_atomic_load(P + fgr - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                        orl = IDS[fg] >> 31; orr = IDS[fgr] >> 31;
                    }
                }
                // overlap test of the step (a lane without a step fails it); then the link is looked for among the (up to four)
                // that sit in the left node's record — straight selects, no branches —; the link table is asked only if it is not
                // there and the node has more links, or for a revisited node (the link between the first occurrences)
                wmask go_m = step_m & m_ge(pre_l, need_l) & m_ge(tot - ((int32_t)fr > (int32_t)lnb ? pre_rx : 0u), need_r);   // (lnb < 0: a sub-pass inside a long line)
                uint32_t hold = 0;                                       // (wave-uniform; a first sweep's sub-pass in front of the line's last: its first link << 8 | links decided — their hits wait in the log)
                if (RARELY(lsub)) {
                    uint32_t adv = n_pass - 1u;                          // how far the line's next sub-pass begins behind this one
                    if ((lsub & L_ONE) && !(lsub & L_FINAL)) {
                        const wmask und_m = step_m & m_ge(pre_l, need_l) & ~go_m;   // the bound did not do: the rest of the path decides
                        if (und_m) adv = (uint32_t)__builtin_ctzll(und_m);
                        if (adv == 0u) {                                 // the sub-pass's first link: sixty-four nodes shorter than the overlap asked for — two sweeps
                            lsub |= L_MEASURE; SET_R(lP);
                            lS += rdlane(gsum, n_pass - 2u); SET_P(lP + n_pass - 1u);
                            continue;
                        }
                        go_m &= low_bits64(adv);
                        hold = (lP << 8) | adv;
                    }
                    if (lsub & L_SWEEP1) go_m &= m_ge(lP + lane, lR);      // (the second sweep: what the first has found stays found)
                    if (!(lsub & L_FINAL)) { lS += rdlane(gsum, adv - 1u); SET_P(lP + adv); }   // (nothing below looks at lS or lP again)
                }
                const uint32_t want = (idr << 2) | orl | (orr << 1);
                const wmask le24 = m_le(len, 24u), le32 = m_le(len, 32u), le40 = m_le(len, 40u);   // (a longer name's bytes sit where the first links would)
                const wmask m0 = le24 & m_eq(r2.x, want), m1 = le32 & m_eq(r2.z, want), m2 = le40 & m_eq(r3.x, want), m3 = m_eq(r3.z, want);
                const wmask inl = m0 | m1 | m2 | m3;
                const uint32_t v = in_mask(m0) ? r2.y : in_mask(m1) ? r2.w : in_mask(m2) ? r3.y : r3.w;
                const wmask found_m = go_m & ~moved_m & inl;
                const wmask ask_m = go_m & (moved_m | ~(inl | m_ne(row_inline, 0u)));
                const bool ask = in_mask(ask_m);
                // hits of the lane: count; bit 31: they are in a list (hp), else in h0 (and h1)
                uint32_t nh = in_mask(found_m) ? 1u : 0u, h0 = v, h1 = 0;
                const uint32_t *hp = nullptr;                             // more than two hits: the list
                {
                    const wmask many = found_m & m_ge(v, 0x80000000u);   // (wave-uniform: a link with several hits)
                    if (RARELY(many)) { if (in_mask(many)) { hp = g.name_ihits + (v & 0x7FFFFFFFu) + 1; nh = hp[-1] | 0x80000000u; } }
                }
                if (RARELY(ask_m) && !DIAG(512u)) {                       // (DIAG 512, measurement only: the link table is never asked)
                    const uint64_t hl = h;                               // (the first occurrence spells the same name)
                    const uint64_t hr = ((uint64_t)lane_above((uint32_t)(h >> 32)) << 32) | lane_above((uint32_t)h);
                    if (ask) {
                        const uint32_t klo = (idr << 1) | orr, khi = (idl << 1) | orl;
                        uint32_t sa, sb2;
                        link_slots(link_prehash(hl, orl, hr, orr), g.link_seed, g.link_mask, sa, sb2);
                        uint4 ek = *(const uint4 *)(g.link_tab + (size_t)sa * 4);
                        const uint4 ek2 = *(const uint4 *)(g.link_tab + (size_t)sb2 * 4);
                        if (!(ek.x == klo && ek.y == khi)) ek = ek2;                          // the other candidate slot
                        if (ek.x == klo && ek.y == khi) {
                            // one hit: (hit, NO_HIT); two: (hit, hit); more: (MANY | index into hits[], count)
                            if ((ek.z & 0x80000000u) && ek.w != 0xFFFFFFFFu && ek.z != 0xFFFFFFFFu) { hp = g.hits + (ek.z & 0x7FFFFFFFu); nh = ek.w | 0x80000000u; }
                            else { h0 = ek.z; h1 = ek.w; nh = ek.w == 0xFFFFFFFFu ? 1u : 2u; }
                        }
                    }
                }
                // A long line's first sweep counts nothing before the line's LAST name is known (r05): a later sub-pass may still hand the
                // whole line to the exact path (a name the table does not hold, a path of 4 Gbp), which counts every link of it.  What the
                // sweep finds on its way waits in the worker's words of global memory, two per link — (0xFFFFFFFF, -): no hit; (hit,
                // 0x7FFFFFFF): one; (hit, hit): two; (offset, 1 << 31 | list << 30 | n): n hits in g.hits / g.name_ihits — and is counted
                // behind the line's last sub-pass (of its only sweep: links [0, lP); of its second sweep: links [0, lR)) by the code that
                // counts any pass's hits, which comes round again for it.  (ONE copy of the counting code, and the log's loads waited for
                // inside their rarely taken block: a flush with code of its own — wherever it stood — or a wait the compiler placed where
                // that block joins the pass cost ordinary text 3 %: profiles/r05/experiments/late_deferral_double_count.txt.)
                if (RARELY(hold)) {                                      // this sub-pass's links [hold >> 8, + hold & 0xFF) -> the log; nothing is counted
                    uint32_t hd = hold, lane_c = lane;
                    asm volatile("" : "+s"(hd), "+v"(lane_c));           // (as below: nothing of this block is to be computed in front of it)
                    uint32_t *LGw = a.long_pre + (size_t)blockIdx.x * LONG_WORDS + LONG_LOG + 2u * ((hd >> 8) + lane_c);
                    uint32_t w0 = 0xFFFFFFFFu, w1 = 0u;
                    if (nh & 0x80000000u) { w0 = (uint32_t)(hp - (ask ? g.hits : g.name_ihits)); w1 = 0x80000000u | (ask ? 0u : 0x40000000u) | (nh & 0x3FFFFFFFu); }   // (a list from the link table / from the left node's record)
                    else if (nh) { w0 = h0; w1 = nh == 1u ? 0x7FFFFFFFu : h1; }
                    if (lane_c < (hd & 0xFFu)) {
                        __hip_atomic_store(LGw, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(LGw + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    nh = 0;
                }
                for (uint32_t fb = 0;;) {
                    // hit records: one aggregated atomic per wave reserves the slots
                    unsigned long long rbase = 0;
                    if (RARELY(a.want_hits)) {
                        uint32_t wtot2, ex = wave_excl_scan(nh & 0x7FFFFFFFu, wtot2);
                        if (wtot2) {
                            if (lane == 0) rbase = atomicAdd(&a.st->n_recs, (unsigned long long)wtot2);
                            rbase = __shfl(rbase, 0) + ex;
                        }
                    }
                    auto emit = [&](uint32_t hv, uint32_t jj) {
                        // counts[slot] = ref | alt << 32 is, in memory, the pair of 32-bit counters [2 * slot + allele] = [hv]: a 32-bit atomic
                        // (the memory-side atomic units take ~14 % more of those per second than 64-bit ones: profiles/r01_ubench_atomics.txt)
                        if (!DIAG(8u)) atomicAdd(&((unsigned int *)a.counts)[DIAG(4u) ? (hv & 1023u) : DIAG(64u) ? ((hv & 63u) | ((blockIdx.x & 1023u) << 6)) : hv], 1u);   // (4 / 64, ablation builds: all updates into 4 KB / into 256 B per worker)
                        if (a.want_hits) {
                            if (rbase + jj < a.rec_cap) {
                                svjg_hitrec r; r.line_start = a.base_offset + c0 + (LINE[lbase + ln] & 0xFFFFu); r.slot = hv >> 1;
                                r.n_ref = (hv & 1u) ? 0 : 1; r.n_alt = (hv & 1u) ? 1 : 0;
                                a.recs[rbase + jj] = r;
                            } else atomicOr(&a.st->overflow, 2u);
                        }
                    };
                    if (in_mask(m_le(nh - 1u, 1u))) emit(h0, 0);        // the usual case: one hit (or two), held in registers
                    if (RARELY(m_gt(nh, 1u))) {
                        const uint32_t n = nh & 0x7FFFFFFFu;
                        for (uint32_t jj = hp ? 0u : 1u; jj < n; ++jj) emit(hp ? hp[jj] : h1, jj);
                    }
                    if (!RARELY(lsub)) break;
                    if (hold) break;
                    // (what follows is a long line's business: its operands are taken through empty asm statements so that the compiler cannot
                    //  compute them in front of the loop, on every pass of ordinary text — it did: +21 scalar instructions a pass)
                    uint32_t ls = lsub, lane_c = lane;
                    asm volatile("" : "+s"(ls), "+v"(lane_c));
                    if (!(ls & L_FINAL)) break;
                    const uint32_t upto = (ls & L_SWEEP1) ? ((ls >> 16) & 0xFFu) : (ls & L_ONE) ? ((ls >> 8) & 0xFFu) : 0u;   // (lR : lP) the line's last sub-pass: the held-back hits of its links [0, upto), 64 links a turn
                    if (fb >= upto) break;
                    if (fb == 0u) long_words_sync();
                    const uint32_t *LG = a.long_pre + (size_t)blockIdx.x * LONG_WORDS + LONG_LOG + 2u * (fb + lane_c);
                    nh = 0; hp = nullptr;
                    if (fb + lane_c < upto) {
                        const uint32_t w0 = __hip_atomic_load(LG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const uint32_t w1 = __hip_atomic_load(LG + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (w0 != 0xFFFFFFFFu) {
                            if (w1 & 0x80000000u) { hp = ((w1 & 0x40000000u) ? g.name_ihits : g.hits) + w0; nh = (w1 & 0x3FFFFFFFu) | 0x80000000u; }
                            else { h0 = w0; h1 = w1; nh = w1 == 0x7FFFFFFFu ? 1u : 2u; }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the log's words are there before the loop comes round — else the compiler puts that wait at the loop's head, where every pass of ordinary text would wait for its predecessors' count updates (3 %)
                    fb += WG;
                }
                tick(6);
                if (RARELY(lsub) && (lsub & L_FINAL)) { ++i0; lsub = 0; }   // (a long line's last sub-pass: on to the next line)
            }
#undef lD0
#undef lP
#undef lR
#undef SET_D0
#undef SET_P
#undef SET_R
            wave_sync();
            // ---- R6: lines for the exact path ------------------------------------------------------------------
            {
                const uint32_t fin = RL[lane].z >> 24;
                const bool defer = lane < cnt && fin >= ST_DEFER;
                unsigned long long db = ballot64(defer);
                if (RARELY(db)) {
#pragma unroll
                    for (uint32_t cse = 0; cse < DC_STRIPE; ++cse) {
                        const unsigned long long cb = ballot64(defer && fin == ST_DEFER + cse);
                        if (cb && lane == 0) atomicAdd(&a.st->cause[cse], (unsigned long long)__popcll(cb));
                    }
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
            __builtin_amdgcn_s_setprio(P_R1);
        }
        wave_lines += n_own;
        wave_sync();                                                     // text, bitmap and lists are overwritten by the next stripe
        tick(7);
        if (last_stripe) break;
        pos = next_pos;
    }
    // the next small chunk (none for inputs that are even shares: 3 584 workers asking one counter at the same moment cost 40 us)
    if (!a.small) break;
    unsigned long long ci = 0;
    if (lane == 0) ci = atomicAdd(&a.st->next_chunk, 1ull);
    ci = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ci >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ci);
    pos = a.begin + (unsigned long long)gridDim.x * a.region + ci * a.small;
    if (pos >= a.n_bytes) break;
    rend = pos + a.small < a.n_bytes ? pos + a.small : a.n_bytes;
  }
    if (lane == 0 && wave_lines) atomicAdd(&a.st->n_lines, wave_lines);
#ifdef SVJG_TIMING
    if ((a.diag & 16u) && lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&a.dbg[i], acc[i]);
#endif
}

// svjg_copy_rate: plain streams, 16 bytes per lane, non-temporal, four loads in flight per lane (what the HBM gives a kernel that does
// nothing else): a copy, and a read that folds what it read into one word per block (written only if it is a value the data cannot make)
typedef uint32_t svjg_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(TPB) void k_copy16(uint4 *dst, const uint4 *src, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * TPB;
    uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const svjg_u32x4 a = __builtin_nontemporal_load((const svjg_u32x4 *)(src + i)), b = __builtin_nontemporal_load((const svjg_u32x4 *)(src + i + stride));
        const svjg_u32x4 c = __builtin_nontemporal_load((const svjg_u32x4 *)(src + i + 2 * stride)), d = __builtin_nontemporal_load((const svjg_u32x4 *)(src + i + 3 * stride));
        __builtin_nontemporal_store(a, (svjg_u32x4 *)(dst + i)); __builtin_nontemporal_store(b, (svjg_u32x4 *)(dst + i + stride));
        __builtin_nontemporal_store(c, (svjg_u32x4 *)(dst + i + 2 * stride)); __builtin_nontemporal_store(d, (svjg_u32x4 *)(dst + i + 3 * stride));
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(TPB) void k_read16(uint4 *sink, const uint4 *src, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * TPB;
    uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const svjg_u32x4 a = __builtin_nontemporal_load((const svjg_u32x4 *)(src + i)), b = __builtin_nontemporal_load((const svjg_u32x4 *)(src + i + stride));
        const svjg_u32x4 c = __builtin_nontemporal_load((const svjg_u32x4 *)(src + i + 2 * stride)), d = __builtin_nontemporal_load((const svjg_u32x4 *)(src + i + 3 * stride));
        acc ^= a.x ^ a.w ^ b.y ^ b.z ^ c.x ^ c.w ^ d.y ^ d.z;
    }
    for (; i < n16; i += stride) acc ^= src[i].x;
    if (acc == 0x12345677u) sink[0].x = acc;
}

// svjg_run_resident: the three things a pass starts from — zero counts (and guard words), a fresh status block, "no row lacked its
// binomial term" — in one launch instead of two memsets and a copy
__global__ __launch_bounds__(TPB) void k_step_reset(unsigned long long *counts, uint64_t n_words, DevStatus *st, unsigned int *max_n) {
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * TPB) counts[i] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        DevStatus z{}; z.err = ~0ull;
        *st = z;
        max_n[0] = 0u; max_n[1] = 0u;
    }
}

// what the exact routine returned for the line at file offset `off`: the exception the reference would die with (the first of
// the file wins), or "the host decides" (svjg.h: SVJG_EXC_ASK_HOST): neither counted nor fatal, its offset is kept
__device__ inline void report_line(const ClassifyArgs &a, unsigned long long off, int rc) {
    if (rc == SVJG_EXC_ASK_HOST) {
        const unsigned long long i = atomicAdd(&a.st->n_host, 1ull);
        if (i < a.host_cap) a.host_lines[i] = off; else atomicOr(&a.st->overflow, 4u);
    } else atomicMin(&a.st->err, (off << 3) | (unsigned long long)rc);
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
#ifndef SVJG_SLOW_LANE_LDS
#define SVJG_SLOW_LANE_LDS (16 * 1024)   /* measured (tools/slowpath_bench.py, 1 M lines): 32 KB 61 ms, 16 KB 30 ms (eight blocks per CU: the register limit), 8 KB 41 ms (lines no longer fit) */
#endif
constexpr uint32_t SLOW_TPB = 64, SLOW_LDS = 32 * 1024, SLOW_MAXLINE = 16 * 1024;
constexpr uint32_t SLOW_LANE_LDS = SVJG_SLOW_LANE_LDS;      // staging buffer of the one-lane-per-line kernel (64 lines: 16 KB holds lines of 256 bytes on average)
// n_def = SLOW_ASK_DEVICE: the launch was enqueued right behind the main kernel without a host round trip (svjg_run_resident); the
// number of deferred lines is what the main kernel left in the status block, and the kernel works only if it lies in (lo, hi]
// (two launches share the range: one wave per line up to a limit, one lane per line beyond it).  A list that overflowed is not
// touched: the host sees the flag and repeats the pass with a larger one.
constexpr uint64_t SLOW_ASK_DEVICE = ~0ull;
__device__ inline uint64_t slow_n_def(const ClassifyArgs &a, uint64_t n_def, uint64_t lo, uint64_t hi) {
    if (n_def != SLOW_ASK_DEVICE) return n_def;
    if (a.st->overflow & 1u) return 0;
    const uint64_t n = a.st->n_deferred;
    return (n > lo && n <= hi) ? n : 0;
}
// (r04) The per-node results — strand of the name's first occurrence, id, get_node_len or the exception it raises — are kept per lane in
// LDS (SLOW_LANE_NODES nodes a line, the lanes of the wave interleaved: no bank conflicts) and every link adds up what is there, as the
// one-wave-per-line kernel does (svjg_line.h: slow_wave_phase1 / phase2 with one lane): O(k) name resolutions a line instead of O(k^2).
// A path of more nodes runs through slow_line as before.
constexpr uint32_t SLOW_LANE_NODES = 12;
__global__ __launch_bounds__(SLOW_TPB) void k_classify_slow(ClassifyArgs a, uint64_t n_def_arg, uint64_t lo, uint64_t hi) {
    const uint64_t n_def = slow_n_def(a, n_def_arg, lo, hi);
    __shared__ __attribute__((aligned(16))) uint8_t stage[SLOW_LANE_LDS];
    __shared__ int64_t c_len[SLOW_LANE_NODES * SLOW_TPB];
    __shared__ uint32_t c_id[SLOW_LANE_NODES * SLOW_TPB];
    __shared__ uint8_t c_rc[SLOW_LANE_NODES * SLOW_TPB], c_strand[SLOW_LANE_NODES * SLOW_TPB];
    const uint32_t lane = threadIdx.x;
#ifdef SVJG_TIMING
    // measurement only (SVJG_DIAG & 16): the longest any block of 64 lines took per step (a.dbg[24 ..]: terminators + staging, per-line part, nodes, links)
    unsigned long long lstamp = __builtin_readcyclecounter();
#define ltick(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); if ((a.diag & 16u) && lane == 0) atomicAdd(&a.dbg[24 + (i)], t_ - lstamp); lstamp = t_; } while (0)
#else
#define ltick(i) do { } while (0)
#endif
    for (uint64_t b0 = (uint64_t)blockIdx.x * SLOW_TPB; b0 < n_def; b0 += (uint64_t)gridDim.x * SLOW_TPB) {
        const bool have = b0 + lane < n_def;
        ltick(4);
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
        const uint32_t want = span <= (SLOW_MAXLINE < SLOW_LANE_LDS ? SLOW_MAXLINE : SLOW_LANE_LDS) ? (uint32_t)span : 0u;
        uint32_t tot;
        const uint32_t off = wave_excl_scan(want, tot);
        const bool staged = have && want && off + want <= SLOW_LANE_LDS;
        if (staged)
            for (uint32_t o = 0; o < want; o += 16) *(uint4 *)(stage + off + o) = *(const uint4 *)(a.gaf + a0 + o);
        __syncthreads();
        ltick(0);
        if (have) {
            // two instances of the string routine: on a pointer the compiler knows to be LDS (ds_read_u8 per byte) and on
            // global memory; one generic pointer would turn every byte access into a flat load (~10x slower per line)
            SlowEmit em{&a, a.base_offset + s};
            int rc;
            if (staged) {
                typedef const __attribute__((address_space(3))) uint8_t *lds_text;
                const lds_text t = (lds_text)(stage + off);
                SlowLine ln;
                rc = slow_prologue(t, s - a0, s - a0 + (e - s), ln);
                ltick(1);
                if (!rc && ln.k >= 2) {
                    if (ln.k <= SLOW_LANE_NODES) {
                        NodeScratch ns{(SVJG_TAB_AS uint32_t *)c_id + lane, (SVJG_TAB_AS int64_t *)c_len + lane, (SVJG_TAB_AS uint8_t *)c_rc + lane, (SVJG_TAB_AS uint8_t *)c_strand + lane, SLOW_LANE_NODES, SLOW_TPB};
                        uint64_t order = 0;
                        rc = slow_wave_phase1(a.g, t, ln, ns, 0u, 1u, &order);
                        ltick(2);
                        if (!rc) rc = slow_wave_phase2(a.g, ln, ns, em, 0u, 1u, &order);
                        ltick(3);
                    } else rc = slow_line(a.g, t, s - a0, s - a0 + (e - s), em);
                }
            } else rc = slow_line(a.g, a.gaf, s, e, em);
            if (rc) report_line(a, a.base_offset + s, rc);
        }
        __syncthreads();
    }
}

// The exact path for a few lines: one WAVE per deferred line.  Every lane runs the per-line part (same control flow in
// all lanes); then lane l takes the path nodes l (mod 64) — strand, id, length or the exception get_node_len raises, kept
// per node in LDS — and after a barrier the links l (mod 64), which only add up what phase 1 left: O(k / 64) name
// resolutions per lane where one lane per line needs O(k^2).  An error is the one the reference would meet first
// (smallest position in its sequence of steps).
constexpr uint32_t SLOW_NODES = 736;                                   // path nodes the per-node scratch of one line holds (28 bytes a node: three blocks' LDS fit a CU)

// which of the eight bytes of w equal the byte c: bit i = byte i (exact)
__device__ inline uint32_t eq_bytes8(uint64_t w, uint32_t c) {
    const uint64_t x = w ^ (0x0101010101010101ull * c);
    const uint64_t z = ~((((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x)) & 0x8080808080808080ull;
    return (uint32_t)(((z >> 7) * 0x0102040810204080ull) >> 56);
}
// svjg_line.h: slow_prologue, shared out over the 64 lanes of the wave that has the line to itself (r04): the three walks over the line
// that one lane makes eight bytes per step — the first twelve tabs, the last "id:f:", the path's nodes (the caller counts them while it
// builds the table of the path's pieces) — take 1 KB per step here; the short parts (rstrip, the nine int() columns, the tag's value) run
// in every lane as before.  Same results, same order of the exceptions.  o.k is left to the caller.
// svjg_line.h: slow_wave_phase2 for a line whose nodes all have a length (no exception waiting in a sum): the two sums of every link come
// from ONE running sum over the path (P[j] = len[0] + ... + len[j], made in place by the caller, wrapping like the sums it replaces) and
// list.index of a link's names (:269-271) from a table of first occurrences (first[j] = the first node with node j's id; the caller fills
// it: j itself where the ids run one way) instead of four walks over the node list per link.
template <class Emit>
__device__ inline int slow_wave_links_summed(const GraphView &g, const SlowLine &ln, const SVJG_TAB_AS uint32_t *id, const SVJG_TAB_AS int64_t *P,
                                             const SVJG_TAB_AS uint8_t *strand, const SVJG_TAB_AS uint32_t *first, Emit &emit, uint32_t lane, uint64_t *order) {
    const uint64_t tot = (uint64_t)P[ln.k - 1];
    for (uint32_t i = lane; i + 1 < ln.k; i += 64) {
        const uint32_t lid = id[i], rid = id[i + 1];
        if (lid == NONE32 || rid == NONE32) continue;
        const uint32_t ei = edge_find(g, lid, strand[i], rid, strand[i + 1]);
        if (ei == NONE32) continue;
        const svjg_edge ed = g.edges[ei];
        const uint32_t nh = ed.meta >> 2;
        if (!nh) continue;
        if (g.dover_list) { *order = (2ull << 32) | i; return SVJG_EXC_TYPE_ERROR; }   // int >= list (:269), before the right sum is formed
        const uint32_t il = first[i], ir = first[i + 1];
        const int64_t left = P[il], right = (int64_t)(tot - (ir ? (uint64_t)P[ir - 1] : 0ull));
        if (left - ln.Ts >= (int64_t)g.d_over && right - (ln.Tlen - ln.Te - 1) >= (int64_t)g.d_over)
            for (uint32_t j = 0; j < nh; ++j) { uint32_t hv = edge_hit(g, ed, j); emit(hv >> 1, hv & 1u); }
    }
    return 0;
}

typedef const __attribute__((address_space(3))) uint8_t *slow_lds_text;
__device__ inline int slow_prologue_wave(slow_lds_text t, uint64_t s, uint64_t e, SlowLine &o, uint32_t lane) {
    o.k = 0;
    while (e > s && py_strip_space(t[e - 1])) --e;
    // the first twelve tabs of t[s, e)
    uint32_t tp[12], nt = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) tp[i] = (uint32_t)e;
    for (uint64_t base = s; base < e && nt < 12; base += 1024) {
        const uint64_t p = base + (uint64_t)lane * 16;
        uint32_t m = 0;
        if (p < e) {
            m = eq_bytes8(ld64(t, p), '\t') | (eq_bytes8(ld64(t, p + 8), '\t') << 8);
            if (p + 16 > e) m &= (1u << (e - p)) - 1u;               // (bytes behind the line's end: whatever the buffer holds)
        }
        for (;;) {
            const unsigned long long hit = __ballot(m != 0);
            if (!hit || nt >= 12) break;
            const int L = __builtin_ctzll(hit);
            const uint32_t pos = (uint32_t)(base + (uint64_t)L * 16) + (uint32_t)__builtin_ctz((uint32_t)__shfl((int)m, L));
#pragma unroll
            for (int i = 0; i < 12; ++i) if ((uint32_t)i == nt) tp[i] = pos;
            ++nt;
            if ((int)lane == L) m &= m - 1u;
        }
    }
    if (nt < 11) return SVJG_EXC_VALUE_ERROR;                         // fewer than twelve fields
    auto fs = [&](int i) -> uint64_t { return i ? (uint64_t)tp[i - 1] + 1 : s; };
    auto fe = [&](int i) -> uint64_t { return (uint64_t)tp[i]; };       // (tp[11] = e when the line has eleven tabs)
    int64_t v6 = 0, v7 = 0, v8 = 0, v10 = 0;
    {
        int64_t v = 0;
#define SVJG_COL(c, keep) do { const int r_ = py_int(t, fs(c), fe(c), v); if (r_ != PY_INT_OK) return r_ == PY_INT_BIG || has_high(t, fs(c), fe(c)) ? SVJG_EXC_ASK_HOST : SVJG_EXC_VALUE_ERROR; keep; } while (0)
        SVJG_COL(1, (void)0); SVJG_COL(2, (void)0); SVJG_COL(3, (void)0); SVJG_COL(6, v6 = v); SVJG_COL(7, v7 = v); SVJG_COL(8, v8 = v);
        SVJG_COL(9, (void)0); SVJG_COL(10, v10 = v); SVJG_COL(11, (void)0);
#undef SVJG_COL
    }
    // the last "id:f:" of the line (:193-196)
    {
        unsigned long long mine = 0;                                  // position + 1 of the last one this lane saw
        for (uint64_t base = s; base + 5 <= e; base += 1024) {
            const uint64_t p = base + (uint64_t)lane * 16;
            if (p + 5 <= e) {
                uint32_t m = eq_bytes8(ld64(t, p), 'i') | (eq_bytes8(ld64(t, p + 8), 'i') << 8);
                if (p + 16 + 4 > e) m &= (1u << (e - 4 - p)) - 1u;     // (an 'i' with fewer than four bytes behind it)
                for (; m; m &= m - 1u) {
                    const uint64_t q = p + (uint32_t)__builtin_ctz(m);
                    if (t[q + 1] == 'd' && t[q + 2] == ':' && t[q + 3] == 'f' && t[q + 4] == ':') mine = q + 1;
                }
            }
        }
#pragma unroll
        for (int d = 32; d; d >>= 1) { const unsigned long long y = __shfl_xor(mine, d); mine = y > mine ? y : mine; }
        if (mine) {
            uint64_t a = mine - 1 + 5, b = a;
            while (b < e && t[b] != '\t') ++b;
            if (!py_float_ok(t, a, b)) return has_high(t, a, b) ? SVJG_EXC_ASK_HOST : SVJG_EXC_VALUE_ERROR;
        } else if (v10 == 0) return SVJG_EXC_ZERO_DIVISION;
    }
    o.ps = fs(5); o.pe = fe(5);
    if (o.pe == o.ps) return SVJG_EXC_INDEX_ERROR;                  // p[0]
    o.oriented = t[o.ps] == '<' || t[o.ps] == '>';
    o.Tlen = v6; o.Ts = v7; o.Te = v8;
    return 0;
}
__global__ __launch_bounds__(SLOW_TPB) void k_classify_slow_wave(ClassifyArgs a, uint64_t n_def_arg, uint64_t lo, uint64_t hi) {
    const uint64_t n_def = slow_n_def(a, n_def_arg, lo, hi);
    __shared__ __attribute__((aligned(16))) uint8_t stage[SLOW_LDS];
    __shared__ int64_t n_len[SLOW_NODES];
    __shared__ uint32_t n_id[SLOW_NODES];
    __shared__ uint8_t n_rc[SLOW_NODES], n_strand[SLOW_NODES];
    __shared__ uint32_t n_piece[SLOW_NODES];                           // the path's pieces: start | length << 16 (svjg_line.h: strand_of_pieces)
    __shared__ uint16_t n_colon[SLOW_NODES];                           // ... where each has its ':' (piece_colons)
    __shared__ uint64_t n_key[SLOW_NODES];                             // ... and what stands around it (piece_key)
    const uint32_t lane = threadIdx.x;
#ifdef SVJG_TIMING
    // measurement only (SVJG_DIAG & 16): the longest any line took per step of this kernel (a.dbg[16 ..]: terminator + staging, per-line part,
    // piece table, nodes, links)
    unsigned long long wstamp = __builtin_readcyclecounter();
#define wtick(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); if ((a.diag & 16u) && lane == 0) atomicMax(&a.dbg[16 + (i)], t_ - wstamp); wstamp = t_; } while (0)
#else
#define wtick(i) do { } while (0)
#endif
    for (uint64_t b = blockIdx.x; b < n_def; b += gridDim.x) {
        const uint64_t s = a.deferred[b];
        wtick(7);
        // the terminator: 64 aligned 16-byte blocks per step
        const uint64_t a0 = s & ~15ull;
        uint64_t e = ~0ull;
        for (uint64_t p0 = a0; e == ~0ull; p0 += 1024) {                 // the buffer is zero padded far beyond n_bytes
            const uint64_t p = p0 + lane * 16;
            const uint4 v = *(const uint4 *)(a.gaf + (p < a.n_bytes + 16 ? p : a0));
            uint32_t m = eq_mask16(v, 0x0A0A0A0Au) | eq_mask16(v, 0x0D0D0D0Du);
            if (p + 16 > a.n_bytes) m |= p >= a.n_bytes ? 0xFFFFu : (0xFFFFu << (a.n_bytes - p)) & 0xFFFFu;   // the text ends here
            if (p < s) m &= 0xFFFFu << (s - p);
            const unsigned long long hit = __ballot(m != 0);
            if (hit) {
                const int first = __builtin_ctzll(hit);
                e = p0 + (uint64_t)first * 16 + (uint64_t)__builtin_ctz((uint32_t)__shfl((int)m, first));
            }
        }
        const uint64_t span = (e - a0 + 15) & ~15ull;                     // bytes of the aligned blocks that hold the line
        const bool staged = span <= SLOW_LDS;
        if (staged) for (uint64_t o = (uint64_t)lane * 16; o < span; o += 1024) *(uint4 *)(stage + o) = *(const uint4 *)(a.gaf + a0 + o);
        __syncthreads();
        wtick(0);
        SlowEmit em{&a, a.base_offset + s};
        uint64_t order = 0;
        int rc = 0;
        auto wave_min = [&](int r, uint64_t ord) {                     // the error the reference meets first, or 0 (same value in every lane)
            unsigned long long key = r ? ((ord << 3) | (unsigned long long)r) : ~0ull;
#pragma unroll
            for (int d = 32; d; d >>= 1) { const unsigned long long y = __shfl_xor(key, d); key = y < key ? y : key; }
            return key == ~0ull ? 0 : (int)(key & 7ull);
        };
        if (staged) {
            typedef const __attribute__((address_space(3))) uint8_t *lds_text;
            const lds_text t = (lds_text)stage;
            SlowLine ln;
            rc = slow_prologue_wave(t, s - a0, s - a0 + (e - s), ln, lane);   // per-line part, shared out over the lanes (the same result in every lane)
            wtick(1);
            if (!rc) {
                // the path's pieces, 64 bytes of the path per step: a piece starts at a byte that is no separator and has one (or the
                // path's start) in front of it.  As many pieces as the path has nodes (extract_nodes keeps the non-empty ones)
                const uint8_t sep1 = ln.oriented ? '<' : ',', sep2 = ln.oriented ? '>' : ',';
                uint32_t run = 0;
                for (uint64_t base = ln.ps; base < ln.pe; base += 64) {
                    const uint64_t q = base + lane;
                    bool st = false;
                    if (q < ln.pe) {
                        const uint8_t c = t[q], pc = q > ln.ps ? (uint8_t)t[q - 1] : sep1;
                        st = c != sep1 && c != sep2 && (pc == sep1 || pc == sep2);
                    }
                    const unsigned long long m = __ballot(st);
                    if (st) { const uint32_t i = run + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); if (i < SLOW_NODES) n_piece[i] = (uint32_t)q; }
                    run += (uint32_t)__popcll(m);
                }
                ln.k = run;
                __syncthreads();
            }
            if (!rc && ln.k >= 2) {
                if (ln.k <= SLOW_NODES) {
                    NodeScratch ns{(SVJG_TAB_AS uint32_t *)n_id, (SVJG_TAB_AS int64_t *)n_len, (SVJG_TAB_AS uint8_t *)n_rc, (SVJG_TAB_AS uint8_t *)n_strand, SLOW_NODES};
                    // the table of the path's pieces: start | length << 16, the length up to the separator(s) in front of the next piece
                    {
                        const uint8_t sep1 = ln.oriented ? '<' : ',', sep2 = ln.oriented ? '>' : ',';
                        for (uint32_t i = lane; i < ln.k; i += 64) {
                            const uint32_t s0 = n_piece[i];
                            uint32_t e0 = i + 1 < ln.k ? n_piece[i + 1] - 1u : (uint32_t)ln.pe;
                            while (e0 > s0 && ((uint8_t)t[e0 - 1] == sep1 || (uint8_t)t[e0 - 1] == sep2)) --e0;
                            n_len[i] = (int64_t)(e0 - s0);                   // (kept aside: the starts are still being read by the neighbours)
                        }
                        __syncthreads();
                        for (uint32_t i = lane; i < ln.k; i += 64) { const uint16_t cw = piece_colons(t, n_piece[i], (uint64_t)n_len[i]); n_colon[i] = cw; n_key[i] = piece_key(t, n_piece[i], cw); n_piece[i] |= (uint32_t)n_len[i] << 16; }
                        __syncthreads();
                    }
                    wtick(2);
                    // every node's id and length first, then the strands: by id where the line allows it (svjg_line.h: slow_wave_strands)
                    slow_wave_resolve(a.g, t, ln, ns, lane, 64u, (const SVJG_TAB_AS uint32_t *)n_piece);
                    __syncthreads();
                    bool clean = ln.oriented, rises = true;
                    for (uint32_t i = lane; i < ln.k; i += 64) {
                        const uint32_t x = n_id[i];
                        clean = clean && slow_node_clean(a.g, x);
                        rises = rises && x != NONE32 && (i == 0 || (n_id[1] > n_id[0] ? x > n_id[i - 1] : x < n_id[i - 1]));
                    }
                    clean = __ballot(!clean) == 0ull;
                    rises = __ballot(!rises) == 0ull;
                    // (the call fills `order`: result and order are separate statements, not two arguments of one call)
                    const int r1 = slow_wave_strands(t, ln, ns, lane, 64u, &order, (const SVJG_TAB_AS uint32_t *)n_piece, (const SVJG_TAB_AS uint16_t *)n_colon, (const SVJG_TAB_AS uint64_t *)n_key, clean, rises);
                    rc = wave_min(r1, order);
                    __syncthreads();
                    wtick(3);
                    if (!rc) {
                        // has every node a length?  Then the links need no walks over the node list: a running sum in place, and a table of
                        // first occurrences where the pieces were (they are not needed any more) — j itself where the ids run one way
                        bool good = true, oneway = true;
                        for (uint32_t i = lane; i < ln.k; i += 64) {
                            const uint32_t x = n_id[i];
                            good = good && n_rc[i] == 0;
                            oneway = oneway && x != NONE32 && (i == 0 || (n_id[1] > n_id[0] ? x > n_id[i - 1] : x < n_id[i - 1]));
                        }
                        if (__ballot(!good) == 0ull) {
                            oneway = __ballot(!oneway) == 0ull;
                            __syncthreads();
                            for (uint32_t j = lane; j < ln.k; j += 64) {
                                uint32_t f = j;
                                if (!oneway) { const uint32_t x = n_id[j]; if (x != NONE32) { f = 0; while (n_id[f] != x) ++f; } }
                                n_piece[j] = f;
                            }
                            // every lane sums a run of consecutive nodes, the runs' totals are scanned across the lanes
                            const uint32_t per = (ln.k + 63u) / 64u, j0 = lane * per < ln.k ? lane * per : ln.k, j1 = j0 + per < ln.k ? j0 + per : ln.k;
                            unsigned long long sum = 0;
                            for (uint32_t j = j0; j < j1; ++j) sum += (unsigned long long)n_len[j];
                            unsigned long long inc = sum;
#pragma unroll
                            for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(inc, d); if ((int)lane >= d) inc += y; }
                            unsigned long long run = inc - sum;
                            for (uint32_t j = j0; j < j1; ++j) { run += (unsigned long long)n_len[j]; n_len[j] = (int64_t)run; }
                            __syncthreads();
                            const int r2 = slow_wave_links_summed(a.g, ln, (const SVJG_TAB_AS uint32_t *)n_id, (const SVJG_TAB_AS int64_t *)n_len, (const SVJG_TAB_AS uint8_t *)n_strand,
                                                                  (const SVJG_TAB_AS uint32_t *)n_piece, em, lane, &order);
                            rc = wave_min(r2, order);
                        } else { const int r2 = slow_wave_phase2(a.g, ln, ns, em, lane, 64u, &order); rc = wave_min(r2, order); }
                    }
                    wtick(4);
                } else { const int r3 = slow_line(a.g, t, s - a0, s - a0 + (e - s), em, lane, 64u, &order); rc = wave_min(r3, order); }   // a path of more nodes than the scratch holds
            }
        } else { const int r4 = slow_line(a.g, a.gaf, s, e, em, lane, 64u, &order); rc = wave_min(r4, order); }
        if (lane == 0 && rc) report_line(a, a.base_offset + s, rc);
        __syncthreads();
    }
}

// Overflow guard of the packed count vector (ref | alt << 32, summed as one 64-bit integer by the kernels' atomics and by the
// all-reduce): the largest ref field and the largest alt field go to the two extra elements behind the vector, which travel
// through the same all-reduce; if the SUM over the ranks of these maxima stays below 2^32 no slot can have carried from one
// half into the other (the host then mirrors the OverflowError of a Python-side sum).
__global__ __launch_bounds__(TPB) void k_counts_guard(unsigned long long *counts, uint32_t n_slots, const DevStatus *st) {
    uint32_t mr = 0, ma = 0;
    for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < n_slots; i += gridDim.x * TPB) {
        const unsigned long long c = counts[i];
        const uint32_t r = (uint32_t)c, al = (uint32_t)(c >> 32);
        mr = r > mr ? r : mr; ma = al > ma ? al : ma;
    }
    for (int d = 32; d; d >>= 1) { const uint32_t y = __shfl_down(mr, d), z = __shfl_down(ma, d); mr = mr > y ? mr : y; ma = ma > z ? ma : z; }
    if ((threadIdx.x & 63) == 0) {
        if (mr) atomicMax(&counts[n_slots + GUARD_MAX_REF], (unsigned long long)mr);
        if (ma) atomicMax(&counts[n_slots + GUARD_MAX_ALT], (unsigned long long)ma);
    }
    // a fused pass under a communicator: "this rank must repeat the pass" travels through the pass's own all-reduce (svjg_pass.h)
    if (st && blockIdx.x == 0 && threadIdx.x == 0) counts[n_slots + GUARD_REPEAT] = pass_repeat_word(st->overflow);
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
    uint8_t *boundary;                 // 1: one of the row's three -10 * (lik + comb) lies within PL_GUARD of an integer (the host recomputes the row like the reference)
    int32_t *pl32;                     // svjg_run_resident: the three PLs as 32-bit integers (nullptr: not wanted); a row whose PLs do not fit sets bit 1 of genotyped[]
    unsigned int *max_n;               // [0] largest n = ref + alt beyond the log10(i!) table, [1] set if a row names a slot >= n_slots
    uint32_t n_slots;
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
    const uint32_t ok = a.ok[r], sl = a.slot[r];
    if (sl != NONE32 && sl >= a.n_slots) { atomicOr(a.max_n + 1, 1u); return false; }   // a caller error, reported after the pass
    if (!(ok & 1u) || sl == NONE32) return false;
    unsigned long long c = a.counts[sl];
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

constexpr double PL_GUARD = 1e-6;

__device__ inline int64_t trunc_dd(dd v) {               // int(Decimal): toward zero
    double t = trunc(v.hi);
    if (t == v.hi) {                                     // hi is integral: the tail decides
        if (v.hi > 0 && v.lo < 0) t -= 1.0;
        else if (v.hi < 0 && v.lo > 0) t += 1.0;
    }
    return (int64_t)t;
}

// (any grid and block size up to TPB: rows in strides of the grid.  svjg_run_begin launches one WAVE per CU: there the kernel runs beside the
//  next pass's classify kernel and is bound by PCIe — its results go straight to pinned host memory.  A single wave of 56 VGPRs fits on a
//  SIMD that holds three classify workers (3 x 128 + 56 <= 512), so the fourteen workers of a CU keep their places whichever kernel
//  arrives first; a 256-thread block needs room on all four SIMDs and displaces a worker)
__global__ __launch_bounds__(TPB) void k_genotype(GenoArgs a) {
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t ref, alt;
    bool go = geno_gate(a, r, ref, alt);
    a.raw[r * 2] = go ? ref : 0; a.raw[r * 2 + 1] = go ? alt : 0;
    a.genotyped[r] = go;
    a.boundary[r] = 0;
    if (!go) { a.gt[r] = 3; a.pl[r * 3] = a.pl[r * 3 + 1] = a.pl[r * 3 + 2] = 0; if (a.pl32) a.pl32[r * 3] = a.pl32[r * 3 + 1] = a.pl32[r * 3 + 2] = 0; continue; }
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
    else atomicMax(a.max_n, n);                          // the log10(i!) table is too short: the host grows it and runs the pass again
    comb = dd{comb.hi, 0.0};                             // the reference rounds log10(comb) to a double first (:313)
    dd ls[3] = {l0, l1, l2};
    bool wide = false, near = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dd s = dd_add(ls[i], comb);
        dd p = dd_add(dd_add(dd_add(s, s), dd_add(s, s)), s);             // 5 s
        p = dd_add(p, p);                                                 // 10 s
        const int64_t v = trunc_dd(dd_neg(p));
        a.pl[r * 3 + i] = v;
        // The reference adds Decimal(math.log10(math.comb(n, k))) (:313): libm's log10 of a big integer rounded to a double, which
        // need not be the correctly rounded value this kernel uses.  The two can differ in the last places; times ten, next to an
        // integer, that could turn a PL by one.  Rows that close are flagged and recomputed on the host (svjg/genotype.py).
        { const double fr = fabs(p.hi - rint(p.hi)); if (fr < PL_GUARD && comb.hi != 0.0) near = true; }   // (comb = log10(1) = 0 on both sides: nothing to disagree about)
        if (a.pl32) { a.pl32[r * 3 + i] = (int32_t)v; if (v != (int64_t)(int32_t)v) wide = true; }
    }
    if (wide) a.genotyped[r] = 3;
    if (near) a.boundary[r] = 1;
  }
}

}  // namespace svjg
