// Micro-benchmark: issue cost of the VALU instruction kinds k_classify_main is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valurate tools/ubench/valurate.hip
// One workgroup of 256 lanes per CU slot, 4 waves per SIMD (the kernel's occupancy); every wave runs N independent
// chains of the instruction under test, so latency is hidden and the figure is SIMD cycles per wave instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 4096
#define CH 8

enum Op { ADD32, XOR32, AND_OR, MUL_LO, MUL_U24, MAD_U24, MAD_U64, MUL_HI, LSHL64, LSHR64, ADD64, BCNT, BCNT64, FF1, FF1_64, DOT4, ALIGNBYTE, PERM, BFE, DPP_ADD, BALLOT, READLANE, SHFL, CMP_CNDMASK, SAD_U8, LSHL_OR, QSAD, PK_MIN, XAD, BITOP3, MUL_LO2, FMA32, NOPS };
static const char *NAMES[] = {"v_add_u32", "v_xor_b32", "v_and_or_b32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_u64_u32", "v_mul_hi_u32", "v_lshlrev_b64", "v_lshrrev_b64", "64-bit add (2 instr)", "v_bcnt_u32_b32", "popcll (2 bcnt)", "v_ffbl_b32 (ctz)", "ctzll", "v_dot4_u32_u8", "v_alignbyte_b32", "v_perm_b32", "v_bfe_u32", "v_add_u32 dpp row_shr", "ballot (v_cmp -> sgpr)", "v_readlane", "ds_bpermute (shfl)", "v_cmp + v_cndmask", "v_sad_u8", "v_lshl_or_b32", "v_qsad_pk_u16_u8", "v_pk_min_u16", "v_xad_u32", "v_bitop3_b32", "v_mul_lo_u32 (by a register)", "v_fma_f32"};

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t seed, uint32_t *out) {
    uint32_t x[CH];
    uint64_t y[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { x[c] = seed * (threadIdx.x + 1) + c * 0x9E3779B9u; y[c] = ((uint64_t)x[c] << 32) | (x[c] * 77u); }
    const uint32_t s = seed | 1u;
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == ADD32) x[c] += s;
            if (OP == XOR32) x[c] ^= s + c;
            if (OP == AND_OR) x[c] = (x[c] & 0x80808080u) | (x[(c + 1) % CH]);
            if (OP == MUL_LO) x[c] *= s;
            if (OP == MUL_U24) x[c] = __umul24(x[c], s);
            if (OP == MAD_U24) x[c] = __umul24(x[c], s) + x[(c + 1) % CH];
            if (OP == MAD_U64) y[c] += (uint64_t)(uint32_t)y[c] * s;
            if (OP == MUL_HI) x[c] = __umulhi(x[c], s);
            if (OP == LSHL64) y[c] = (y[c] << (x[c] & 63u)) | 1u;
            if (OP == LSHR64) y[c] = (y[c] >> (x[c] & 63u)) | (1ull << 63);
            if (OP == ADD64) y[c] += ((uint64_t)s << 32) | s;
            if (OP == BCNT) x[c] = __popc(x[c]) + x[c];
            if (OP == BCNT64) y[c] += __popcll(y[c]);
            if (OP == FF1) x[c] += __builtin_ctz(x[c] | 0x80000000u);
            if (OP == FF1_64) y[c] += __builtin_ctzll(y[c] | (1ull << 63));
            if (OP == DOT4) x[c] = __builtin_amdgcn_udot4(x[c], 0x08040201u, x[(c + 1) % CH], false);
            if (OP == ALIGNBYTE) x[c] = __builtin_amdgcn_alignbyte(x[c], x[(c + 1) % CH], s);
            if (OP == PERM) x[c] = __builtin_amdgcn_perm(x[c], x[(c + 1) % CH], 0x07020500u);
            if (OP == BFE) x[c] = __builtin_amdgcn_ubfe(x[c], 3, 17) + s;
            if (OP == DPP_ADD) x[c] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[c], 0x111, 0xF, 0xF, false);
            if (OP == BALLOT) x[c] += (uint32_t)__ballot(x[c] > s);
            if (OP == READLANE) x[c] += (uint32_t)__builtin_amdgcn_readlane((int)x[c], 63);
            if (OP == SHFL) x[c] = (uint32_t)__shfl((int)x[c], (int)(x[c] & 63u)) + 1u;
            if (OP == CMP_CNDMASK) x[c] = x[c] > s ? x[c] - s : x[(c + 1) % CH];
            if (OP == SAD_U8) x[c] = __builtin_amdgcn_sad_u8(x[c], s, x[(c + 1) % CH]);
            if (OP == LSHL_OR) x[c] = (x[c] << 3) | x[(c + 1) % CH];
            if (OP == QSAD) y[c] = __builtin_amdgcn_qsad_pk_u16_u8(y[c], s, y[(c + 1) % CH]);
            if (OP == PK_MIN) { typedef uint16_t u16x2 __attribute__((ext_vector_type(2))); x[c] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x[c]), __builtin_bit_cast(u16x2, x[(c + 1) % CH] + s))); }
            if (OP == XAD) { uint32_t r; asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x[c]), "s"(s), "v"(x[(c + 1) % CH])); x[c] = r; }
            if (OP == BITOP3) x[c] = (x[c] & ~x[(c + 1) % CH]) ^ (x[(c + 2) % CH] | s);
            if (OP == MUL_LO2) x[c] = x[c] * x[(c + 1) % CH];
            if (OP == FMA32) { float f = __builtin_bit_cast(float, x[c]); f = __builtin_fmaf(f, 1.0001f, 0.5f); x[c] = __builtin_bit_cast(uint32_t, f); }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc ^= x[c] ^ (uint32_t)y[c] ^ (uint32_t)(y[c] >> 32);
    if (acc == 0x12345678u) out[0] = acc;
}

template <int OP>
void run(uint32_t *out, double clock_ghz) {
    const int blocks = 256 * 4;                                          // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, 3u, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, 5u, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double per_simd = 4.0 * REP * CH;                              // wave "operations" (source-level) per SIMD
    printf("%-26s %8.3f ms   %6.2f SIMD cycles per wave operation\n", NAMES[OP], ms, ms * 1e-3 * clock_ghz * 1e9 / per_simd);
}

template <int OP> void all(uint32_t *out, double ghz) { run<OP>(out, ghz); if constexpr (OP + 1 < NOPS) all<OP + 1>(out, ghz); }

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    printf("%s, %d CUs, %.2f GHz (nominal; cycles below assume it)\n", p.name, p.multiProcessorCount, ghz);
    uint32_t *out; hipMalloc(&out, 4);
    all<0>(out, ghz);
    return 0;
}
