// Micro-benchmark: do the cache-policy bits of global_atomic_add change where a no-return 32-bit atomic is carried out (MI355X)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/atomics2 tools/ubench/atomics2.hip
// Variants: plain, sc1, nt; addresses random in a table, or random in a copy of the table
// private to the XCD the wave runs on (XCC_ID), or in LDS (ds_add_u32, the ceiling).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ inline uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int MODE>
__global__ void k(unsigned int *tab, uint64_t n_slots, uint64_t per_thread) {
    __shared__ unsigned int lds[8192];
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int xcc = 0;
    if (MODE == 5) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7u; }
    unsigned int one = 1u;
    for (uint64_t i = 0; i < per_thread; ++i) {
        uint64_t r = mix(gid * 1315423911ull + i);
        unsigned int *p = tab + (r % n_slots);
        if (MODE == 0) asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(one) : "memory");
        // (sc0 = "return the old value": not measured — an inline-asm load result arrives asynchronously, behind the compiler's back)
        if (MODE == 2) asm volatile("global_atomic_add %0, %1, off sc1" :: "v"(p), "v"(one) : "memory");
        if (MODE == 3) asm volatile("global_atomic_add %0, %1, off nt" :: "v"(p), "v"(one) : "memory");
        if (MODE == 4) atomicAdd(&lds[r & 8191u], 1u);
        if (MODE == 5) { p = tab + (uint64_t)xcc * n_slots + (r % n_slots); asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(one) : "memory"); }
        if (MODE == 6) { unsigned int v = __builtin_nontemporal_load(p); __builtin_nontemporal_store(v + 1u, p); }   // (not atomic: what a plain read-modify-write costs)
    }
    if (MODE == 4) { __syncthreads(); if (lds[threadIdx.x] == 0xFFFFFFFFu) tab[0] = 0; }
}

template <int MODE>
float run(unsigned int *d, uint64_t n_slots, uint64_t per_thread, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, n_slots, per_thread / 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, n_slots, per_thread);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const int blocks = 2048; const uint64_t per_thread = 64;
    const double total = (double)blocks * 256 * per_thread;
    const char *names[] = {"plain", "(skipped)", "sc1", "nt", "LDS ds_add_u32", "plain, per-XCD copy", "load + store (not atomic)"};
    for (uint64_t n_slots : {32768ull, 262144ull, 2097152ull}) {
        unsigned int *d; hipMalloc(&d, n_slots * 4 * 8); hipMemset(d, 0, n_slots * 4 * 8);
        float ms[7];
        ms[0] = run<0>(d, n_slots, per_thread, blocks); ms[1] = 0;
        ms[2] = run<2>(d, n_slots, per_thread, blocks); ms[3] = run<3>(d, n_slots, per_thread, blocks);
        ms[4] = run<4>(d, n_slots, per_thread, blocks); ms[5] = run<5>(d, n_slots, per_thread, blocks); ms[6] = run<6>(d, n_slots, per_thread, blocks);
        for (int m = 0; m < 7; ++m)
            printf("slots %9llu (%7.1f KB)  %-28s %8.3f ms  %7.2f G ops/s\n", (unsigned long long)n_slots, n_slots * 4 / 1024.0, names[m], ms[m], total / ms[m] / 1e6);
        hipFree(d);
    }
    return 0;
}
