// Byte classes of a 64-byte span of GAF text by BIT PLANES (phase B1 of k_classify_main).
//
// The sixteen words of a span are transposed in registers into eight planes of 64 bits (bit i of plane k = bit k of byte i);
// a byte class is then a boolean function of the planes that handles 32 bytes per instruction and leaves its result as
// the position-ordered bit mask the later phases want — no per-class gather.  Cost: 8 instructions per word for the transposition
// (two v_perm_b32 + three two-register bit exchanges of v_lshl / v_bfi pairs), ~1.3 per word for all classes together; the
// SWAR-per-class version it replaces needed ~22 per word (five v_xad, their flag masks, four v_dot4 gathers, the "id" / "d:"
// halfword minima).  There is no ASCII special case either: a byte >= 0x80 has plane 7 set and is in no class.
//
// What is classified (reference: /root/reference/filter-alignments.py:123-198 reads the file in text mode and splits at tabs):
//   '\n', '\r' (Python's universal newlines), '\t', '<' '>' (path orientation marks, extract_nodes :351-373), decimal digits
//   (int() of the columns, :186-192) and the byte pair "d:" (only the optional "id:f:" tag can make a line's outcome depend on its
//   tags, :193-196; every such tag holds the pair).
//
// Plain C++ so that tests/hostsim can check it against a byte loop on the CPU (test harness only).
#pragma once
#include <stdint.h>

#ifndef SVJG_HD
#define SVJG_HD __host__ __device__ inline __attribute__((always_inline))
#endif

namespace svjg {

// D.byte[i] = byte sel.byte[i] of the eight bytes {hi, lo} (lo = bytes 0..3); selectors 0..7 only
SVJG_HD uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= (uint32_t)((v >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
    return r;
#endif
}

// (m & x) | (~m & y).  On the device as the one instruction it is: left to itself the compiler regroups the masks of consecutive
// exchanges and ends up with a third more instructions.
SVJG_HD uint32_t bfi_b32(uint32_t m, uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(m), "v"(x), "v"(y));
    return r;
#else
    return (m & x) | (~m & y);
#endif
}
// any boolean function of three words: E is written on A, B, C (v_bitop3_b32; its table is E evaluated on 0xF0, 0xCC, 0xAA)
#if defined(__HIP_DEVICE_COMPILE__)
#define SVJG_B3(x, y, z, E) ([&]() -> uint32_t { constexpr uint32_t A = 0xF0u, B = 0xCCu, C = 0xAAu; constexpr uint32_t tbl = (uint32_t)(E) & 0xFFu; \
                                                 return (uint32_t)__builtin_amdgcn_bitop3_b32((x), (y), (z), tbl); }())
#else
#define SVJG_B3(x, y, z, E) ([&]() -> uint32_t { const uint32_t A = (x), B = (y), C = (z); (void)A; (void)B; (void)C; return (uint32_t)(E); }())
#endif

// two registers trade the index bit "register a or b" for the index bit "bit position & D" (M = positions with that bit clear)
template <uint32_t D, uint32_t M>
SVJG_HD void bit_exchange(uint32_t &a, uint32_t &b) {
    const uint32_t na = bfi_b32(M, a, b << D);
    const uint32_t nb = bfi_b32(M, a >> D, b);
    a = na; b = nb;
}

// w[j] = bytes 4j .. 4j+3 of the span (little endian)  ->  w[(i5 * 4 + (k & 3)) * 2 + (k >> 2)] = plane k of bytes 32 * i5 .. + 31
// Index algebra: a bit of the span is (i5 i4 i3 i2 | i1 i0 | k2 k1 k0) = (register | byte in it | bit in it).  The byte transposes
// trade (i4 i3) for (i1 i0), the three exchanges trade k2 k1 k0 for i2 i1 i0: (i5 k1 k0 k2 | i4 i3 | i2 i1 i0).
SVJG_HD void span_planes(uint32_t w[16]) {
#pragma unroll
    for (uint32_t i5 = 0; i5 < 2; ++i5) {
        uint32_t o[4][2];
#pragma unroll
        for (uint32_t i2 = 0; i2 < 2; ++i2) {
            const uint32_t a = w[i5 * 8 + 0 + i2], b = w[i5 * 8 + 2 + i2], c = w[i5 * 8 + 4 + i2], d = w[i5 * 8 + 6 + i2];
            const uint32_t x0 = perm_b32(b, a, 0x05010400u), x1 = perm_b32(b, a, 0x07030602u);   // a0 b0 a1 b1 | a2 b2 a3 b3
            const uint32_t y0 = perm_b32(d, c, 0x05010400u), y1 = perm_b32(d, c, 0x07030602u);
            o[0][i2] = perm_b32(y0, x0, 0x05040100u); o[1][i2] = perm_b32(y0, x0, 0x07060302u);  // a_t b_t c_t d_t
            o[2][i2] = perm_b32(y1, x1, 0x05040100u); o[3][i2] = perm_b32(y1, x1, 0x07060302u);
        }
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) bit_exchange<4, 0x0F0F0F0Fu>(o[t][0], o[t][1]);
#pragma unroll
        for (uint32_t s = 0; s < 2; ++s) {
            bit_exchange<2, 0x33333333u>(o[0][s], o[2][s]); bit_exchange<2, 0x33333333u>(o[1][s], o[3][s]);
            bit_exchange<1, 0x55555555u>(o[0][s], o[1][s]); bit_exchange<1, 0x55555555u>(o[2][s], o[3][s]);
        }
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) { w[(i5 * 4 + t) * 2 + 0] = o[t][0]; w[(i5 * 4 + t) * 2 + 1] = o[t][1]; }
    }
}

struct HalfClasses {                                   // one bit per byte of 32 bytes
    uint32_t nl, cr, tab, ori, nd, dee, colon, high;   // '\n' | '\r' | '\t' | '<' '>' | neither digit nor tab | 'd' | ':' | >= 0x80
};

// p = the eight planes of 32 bytes (p[k & 3][k >> 2] as span_planes leaves them: pass w + i5 * 8)
SVJG_HD HalfClasses half_classes(const uint32_t *p) {
    const uint32_t P0 = p[0], P1 = p[2], P2 = p[4], P3 = p[6], P4 = p[1], P5 = p[3], P6 = p[5], P7 = p[7];
    HalfClasses r;
    const uint32_t n765 = SVJG_B3(P7, P6, P5, ~(A | B | C));
    const uint32_t c0 = SVJG_B3(n765, P4, P3, A & ~B & C);              // 0x08 .. 0x0F
    const uint32_t c00 = c0 & ~P2;                                      // 0x08 .. 0x0B
    r.tab = SVJG_B3(c00, P1, P0, A & ~B & C);                           // 0x09
    r.nl = SVJG_B3(c00, P1, P0, A & B & ~C);                            // 0x0A
    r.cr = SVJG_B3(c0, P2, P1, A & B & ~C) & P0;                        // 0x0D
    const uint32_t h3 = SVJG_B3(P7, P6, P5, ~A & ~B & C);               // 0x20 .. 0x3F
    const uint32_t d1 = SVJG_B3(P3, P2, P1, ~(A & (B | C)));
    const uint32_t dig = SVJG_B3(h3, P4, d1, A & B & C);                // 0x30 .. 0x39
    const uint32_t g0 = SVJG_B3(h3, P4, P3, A & B & C) & ~P0;           // 0x38, 0x3A, 0x3C, 0x3E
    r.ori = g0 & P2;                                                    // 0x3C, 0x3E
    r.colon = SVJG_B3(g0, P2, P1, A & ~B & C);                          // 0x3A
    r.nd = SVJG_B3(dig, r.tab, r.tab, ~(A | B));
    const uint32_t e6 = SVJG_B3(P7, P6, P5, ~A & B & C);                // 0x60 .. 0x7F
    const uint32_t e60 = SVJG_B3(e6, P4, P3, A & ~B & ~C);              // 0x60 .. 0x67
    r.dee = SVJG_B3(e60, P2, P1, A & B & ~C) & ~P0;                     // 0x64
    r.high = P7;
    return r;
}

}  // namespace svjg
