// Micro-benchmark: rate of scattered 64-bit integer atomic adds on MI355X (sizing the count-vector commit).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/atomics tools/ubench/atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ inline uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int MODE>
__global__ void k(unsigned long long *tab, uint64_t n_slots, uint64_t per_thread) {
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = 0; i < per_thread; ++i) {
        uint64_t r = mix(gid * 1315423911ull + i);
        uint64_t s = r % n_slots;
        if (MODE == 0) atomicAdd(&tab[s], 1ull);                                         // agent scope, random slot
        else if (MODE == 1) __hip_atomic_fetch_add(&tab[s], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) { s = (s & ~1ull) | (threadIdx.x & 1); atomicAdd(&tab[s], 1ull); }   // lane pairs share a 16-B span
        else if (MODE == 3) tab[s] = r;                                                 // plain scattered store (reference rate)
        else if (MODE == 4) { s = ((r % (n_slots / 8)) * 8) | (threadIdx.x & 7); atomicAdd(&tab[s], 1ull); } // 8 lanes share a 64-B line
        else if (MODE == 5) atomicAdd((unsigned int *)&tab[s], 1u);                      // 32-bit
    }
}

template <int MODE>
float run(unsigned long long *d, uint64_t n_slots, uint64_t per_thread, int blocks) {
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
    const char *names[] = {"agent u64 random", "workgroup-scope u64 random", "agent u64 lane-pairs adjacent", "plain store u64 random",
                           "agent u64 8 lanes per 64B line", "agent u32 random"};
    for (uint64_t n_slots : {16384ull, 131072ull, 1048576ull, 8388608ull}) {
        unsigned long long *d; hipMalloc(&d, n_slots * 8); hipMemset(d, 0, n_slots * 8);
        float ms[6];
        ms[0] = run<0>(d, n_slots, per_thread, blocks); ms[1] = run<1>(d, n_slots, per_thread, blocks);
        ms[2] = run<2>(d, n_slots, per_thread, blocks); ms[3] = run<3>(d, n_slots, per_thread, blocks);
        ms[4] = run<4>(d, n_slots, per_thread, blocks); ms[5] = run<5>(d, n_slots, per_thread, blocks);
        for (int m = 0; m < 6; ++m)
            printf("slots %9llu (%7.1f KB)  %-34s %8.3f ms  %7.2f G ops/s\n", (unsigned long long)n_slots, n_slots * 8 / 1024.0, names[m], ms[m], total / ms[m] / 1e6);
        hipFree(d);
    }
    return 0;
}
