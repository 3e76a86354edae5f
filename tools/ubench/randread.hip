// Micro-benchmark: rate and latency of random 16-byte loads from a table (sizing the node-name / link table probes).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/randread tools/ubench/randread.hip
// Each lane issues DEPTH independent 16-byte loads per round (all in flight together), then consumes them; rounds are
// dependent (the next addresses come from the loaded data), like the probe -> compare -> next pass chain of the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ inline uint32_t mix(uint32_t z) { z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16; return z; }

template <int DEPTH, int PAIR>
__global__ __launch_bounds__(512) void k(const uint4 *tab, uint32_t mask, uint32_t rounds, uint32_t *out) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = mix(gid * 2654435761u + 12345u), acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint4 v[DEPTH][PAIR];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            uint32_t slot = mix(s + d * 0x9E3779B9u) & mask;       // 64-byte entry
#pragma unroll
            for (int p = 0; p < PAIR; ++p) v[d][p] = tab[(size_t)slot * 4 + p];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int p = 0; p < PAIR; ++p) acc += v[d][p].x ^ v[d][p].w;
        s = mix(s ^ acc);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH, int PAIR>
void run(const uint4 *d, uint32_t mask, int blocks, const char *name, double mb) {
    uint32_t *out; hipMalloc(&out, 4);
    const uint32_t rounds = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<DEPTH, PAIR>), dim3(blocks), dim3(512), 0, 0, d, mask, 8u, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<DEPTH, PAIR>), dim3(blocks), dim3(512), 0, 0, d, mask, rounds, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lines = (double)blocks * 512 * rounds * DEPTH;
    printf("table %7.1f MB  blocks %4d  %-28s %7.3f ms  %7.2f G lines/s  %6.2f TB/s(64B)  round %6.2f us\n", mb, blocks, name, ms, lines / ms / 1e6,
           lines * 64 / ms / 1e9, ms * 1e3 / rounds);
    hipFree(out);
}

int main() {
    for (uint64_t entries : {1ull << 14, 1ull << 17, 1ull << 19, 1ull << 22}) {          // 1 MB, 8 MB, 32 MB, 256 MB
        uint4 *d; hipMalloc(&d, entries * 64); hipMemset(d, 1, entries * 64);
        double mb = entries * 64 / 1048576.0;
        for (int blocks : {256, 512, 1024}) {                                          // 8, 16, 32 waves per CU
            run<1, 1>(d, entries - 1, blocks, "1 line, 1x16B per lane", mb);
            run<2, 2>(d, entries - 1, blocks, "2 lines, 2x16B each", mb);
            run<4, 2>(d, entries - 1, blocks, "4 lines, 2x16B each", mb);
            run<8, 1>(d, entries - 1, blocks, "8 lines, 1x16B each", mb);
        }
        hipFree(d);
    }
    return 0;
}
