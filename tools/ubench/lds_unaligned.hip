// Does the LDS of gfx950 serve ds_read_b32 / b64 / b128 at addresses that are not multiples of the access size?  (measurement
// only: decides whether the node pass may read name bytes without v_alignbyte)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_unaligned tools/ubench/lds_unaligned.hip && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

__global__ void k(uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
    for (uint32_t i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t off = threadIdx.x * 5 + 1;                            // every residue mod 4, 8, 16
    const uint32_t addr = (uint32_t)(uintptr_t)lds + off;
    uint32_t a; uint64_t b; uint32_t c0, c1, c2, c3;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 c;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(addr));
    c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
    uint32_t *o = out + threadIdx.x * 8;
    o[0] = off; o[1] = a; o[2] = (uint32_t)b; o[3] = (uint32_t)(b >> 32); o[4] = c0; o[5] = c1; o[6] = c2; o[7] = c3;
}

int main() {
    uint32_t *d; hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[64 * 8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    uint8_t ref[1024]; for (int i = 0; i < 1024; ++i) ref[i] = (uint8_t)(i * 7 + 3);
    int bad32 = 0, bad64 = 0, bad128 = 0;
    for (int t = 0; t < 64; ++t) {
        const uint32_t off = h[t * 8];
        if (memcmp(&h[t * 8 + 1], ref + off, 4)) ++bad32;
        if (memcmp(&h[t * 8 + 2], ref + off, 8)) ++bad64;
        if (memcmp(&h[t * 8 + 4], ref + off, 16)) ++bad128;
    }
    printf("unaligned LDS reads, 64 offsets: b32 wrong %d, b64 wrong %d, b128 wrong %d\n", bad32, bad64, bad128);
    return 0;
}
