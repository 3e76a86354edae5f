// measurement only — DESIGN.md §10: the memory side of the TWO-KERNEL division of the work that was costed there and not built.
//   K1 streams the text like the main kernel does (persistent single-wave workers, 8 KB stripes through LDS, eight non-temporal 16-byte
//      loads per lane and stripe) and WRITES a compact node stream, 8 bytes per path node (46.6 M nodes at configs[2]: 373 MB), nothing else;
//   K2 is made of light waves (no LDS, few registers: twenty and more per CU beside K1's workers): every lane takes one node of the stream
//      at a time — 8 bytes read, coalesced —, then a 2-byte displacement load at an address that depends on what it read, then the four
//      16-byte loads of one random 64-byte record, then, for 76 % of the nodes, one 32-bit no-return atomic add into the count vector
//      (the real stream of count updates: tools/sol_hits.py, or uniform counters).
// Each alone, and both at once on two streams (in the real pipeline K2 would work on the stream K1 wrote for the pass before).  No byte
// work, no name compares: what the memory system gives this division, to hold against the 0.8 ms it was sketched for.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/split tools/ubench/split.hip ; tools/ubench/split c3|c4shard [reps] [hits file] [K1 workers per CU] [K2 waves per CU]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t WG = 64, TEXT = 8192, HALF = 4096, LDS_WORKER = 11520;

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
__device__ inline uint32_t mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

struct A1 { const uint8_t *text; uint64_t n_bytes, region, small; unsigned long long *next_chunk; uint2 *stream; uint32_t nodes_per_stripe; unsigned long long *sink; };

__global__ __launch_bounds__(WG, 4) void k1_text(A1 a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x;
    uint64_t pos = (uint64_t)blockIdx.x * a.region;
    uint64_t end = pos + a.region < a.n_bytes ? pos + a.region : a.n_bytes;
    if (pos >= end) return;
    uint32_t acc = 0;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint4 pf[4];
    auto fetch_half = [&](uint64_t at) {
        const uint4 *src = (const uint4 *)(a.text + at) + lane;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) { const u32x4 v = __builtin_nontemporal_load((const u32x4 *)(src + i * WG)); pf[i] = make_uint4(v.x, v.y, v.z, v.w); }
    };
    for (;;) {
        fetch_half(pos);
        for (; pos < end; pos += TEXT) {
#pragma unroll
            for (uint32_t h = 0; h < 2; ++h) {
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) *(uint4 *)(lds + h * HALF + (i * WG + lane) * 16) = pf[i];
                if (h == 0) fetch_half(pos + HALF);
                else if (pos + TEXT < end) fetch_half(pos + TEXT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                acc ^= *(const uint32_t *)(lds + h * HALF + ((lane * 68u) & (HALF - 4u)));
            }
            // the stripe's nodes: 8 bytes each, written coalesced at the stripe's place in the stream
            uint2 *out = a.stream + (pos / TEXT) * a.nodes_per_stripe;
            for (uint32_t j = lane; j < a.nodes_per_stripe; j += WG) out[j] = make_uint2(mix((uint32_t)(pos >> 13) * 0x9E3779B1u + j) ^ (acc & 0u), acc | 1u);
        }
        if (!a.small) break;
        unsigned long long ci = 0;
        if (lane == 0) ci = atomicAdd(a.next_chunk, 1ull);
        ci = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ci >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ci);
        pos = (uint64_t)gridDim.x * a.region + ci * a.small;
        if (pos >= a.n_bytes) break;
        end = pos + a.small < a.n_bytes ? pos + a.small : a.n_bytes;
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

struct A2 { const uint2 *stream; uint64_t n_nodes; const uint16_t *disp; uint32_t n_buckets; const uint4 *recs; uint32_t n_recs;
            unsigned int *counts; uint32_t n_counters; const uint32_t *hits; uint64_t n_hits; uint32_t hit_frac; uint32_t zero; uint32_t mode; unsigned long long *sink; };

__global__ __launch_bounds__(256) void k2_tables(A2 a) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_nodes; i += stride) {
        const uint2 nd = a.stream[i];                                        // hash of the name (stands in for it), line / position word
        const uint32_t d = a.disp[mulhi(nd.x, a.n_buckets)];
        const uint32_t slot = mulhi(mix(nd.x ^ 0x5bd1e995u), a.n_recs) + (d & a.zero);
        const uint4 *e = a.recs + (size_t)slot * 4;
        if (a.mode & 1u) { const uint4 r0 = e[0], r1 = e[1]; acc ^= r0.x ^ r0.w ^ r1.z; }              // (a 32-byte record: two loads)
        else { const uint4 r0 = e[0], r1 = e[1], r2 = e[2], r3 = e[3]; acc ^= r0.x ^ r0.w ^ r1.z ^ r2.x ^ r2.z ^ r3.x ^ r3.z; }
        if ((mix(nd.x + 77u) & 1023u) < a.hit_frac) {
            const uint32_t hv = a.hits ? a.hits[(i * 3 / 4) % a.n_hits] : mulhi(mix(nd.x + 0x27D4EB2Fu), a.n_counters);
            atomicAdd(&a.counts[hv + (acc & a.zero)], 1u);                  // (after the record has arrived, as in the real kernel)
        }
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

// K3: "follow the path" with light waves — one ALIGNMENT per lane at a time: 8 bytes of a per-alignment stream read, the FIRST node through
// the displacement array and its record, every further node of the path through its predecessor's record (records in walk order: the
// next record lies right behind; the address still comes out of the record just read: a chain of k dependent 64-byte reads), a count
// update for 76 % of the nodes.  k = 1 + (hash & 7) clipped so that the mean is the workload's nodes per alignment.
struct A3 { const uint2 *stream; uint64_t n_aln; const uint16_t *disp; uint32_t n_buckets; const uint4 *recs; uint32_t n_recs;
            unsigned int *counts; uint32_t n_counters; const uint32_t *hits; uint64_t n_hits; uint32_t hit_frac; uint32_t zero; uint32_t k_mean16; unsigned long long *sink; };
__global__ __launch_bounds__(256) void k3_follow(A3 a) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_aln; i += stride) {
        const uint2 al = a.stream[i];
        const uint32_t d = a.disp[mulhi(al.x, a.n_buckets)];
        uint32_t slot = mulhi(mix(al.x ^ 0x5bd1e995u), a.n_recs - 16u) + (d & a.zero);
        const uint32_t k = 1u + ((mix(al.x + 9u) % 16u) * a.k_mean16 >> 7);     // 1 .. ~2 * mean - 1
        for (uint32_t j = 0; j < k; ++j) {
            const uint4 *e = a.recs + (size_t)slot * 4;
            const uint4 r0 = e[0], r1 = e[1], r2 = e[2], r3 = e[3];
            const uint32_t x = r0.x ^ r0.w ^ r1.z ^ r2.x ^ r2.z ^ r3.x ^ r3.z;
            acc ^= x;
            if ((mix(al.x + 77u + j) & 1023u) < a.hit_frac) {
                const uint32_t hv = a.hits ? a.hits[(i * 7 / 2 + j) % a.n_hits] : mulhi(mix(al.x + 0x27D4EB2Fu + j), a.n_counters);
                atomicAdd(&a.counts[hv + (x & a.zero)], 1u);
            }
            slot = slot + 1u + (x & a.zero);                                 // the successor: named by the record just read
        }
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

struct Workload { const char *name; uint64_t text; uint32_t rec_slots, buckets, count_slots; double nodes, hits, aln; };
static const Workload WL[] = {
    {"c3", 2133165175ull, 247631, 24763, 104881, 46.63e6, 35.53e6, 10e6},  // (displacement buckets at eight names each, as shipped)
    {"c4shard", 2693024596ull, 1237826, 123783, 524826, 58.22e6, 44.34e6, 12.5e6},
};

int main(int argc, char **argv) {
    const char *name = argc > 1 ? argv[1] : "c3";
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const char *hits_path = argc > 3 && strcmp(argv[3], "-") ? argv[3] : nullptr;
    const int w1 = argc > 4 ? atoi(argv[4]) : 8, w2 = argc > 5 ? atoi(argv[5]) : 20;
    const Workload *w = nullptr;
    for (const Workload &x : WL) if (!strcmp(x.name, name)) w = &x;
    if (!w) { fprintf(stderr, "workload c3 | c4shard\n"); return 1; }
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const uint64_t n_stripes = (w->text + TEXT - 1) / TEXT;
    const uint32_t nps = (uint32_t)(w->nodes / (double)n_stripes + 0.5);
    const uint64_t n_nodes = n_stripes * nps;
    uint8_t *text; CHECK(hipMalloc(&text, w->text + 2 * TEXT + 64)); CHECK(hipMemset(text, 0x41, w->text + 2 * TEXT + 64));
    uint2 *stream_a, *stream_b;                                            // K1 writes one, K2 reads the other (the pass before)
    CHECK(hipMalloc(&stream_a, (n_nodes + 256) * 8)); CHECK(hipMalloc(&stream_b, (n_nodes + 256) * 8));
    { std::vector<uint2> h(n_nodes); uint32_t x = 12345u; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = make_uint2(x ^ (x >> 15), 1u); }
      CHECK(hipMemcpy(stream_b, h.data(), n_nodes * 8, hipMemcpyHostToDevice)); }
    uint16_t *disp; CHECK(hipMalloc(&disp, (size_t)w->buckets * 2)); CHECK(hipMemset(disp, 0, (size_t)w->buckets * 2));
    uint4 *recs; CHECK(hipMalloc(&recs, (size_t)w->rec_slots * 64)); CHECK(hipMemset(recs, 0x5A, (size_t)w->rec_slots * 64));
    unsigned int *counts; CHECK(hipMalloc(&counts, (size_t)w->count_slots * 8)); CHECK(hipMemset(counts, 0, (size_t)w->count_slots * 8));
    unsigned long long *sink; CHECK(hipMalloc(&sink, 8));
    A1 a1{}; a1.text = text; a1.n_bytes = w->text; a1.stream = stream_a; a1.nodes_per_stripe = nps; a1.sink = sink;
    const uint32_t grid1 = (uint32_t)(n_cu * w1);
    a1.region = (uint64_t)((double)((w->text + grid1 - 1) / grid1) * 0.85) / TEXT * TEXT; a1.small = 65536;
    CHECK(hipMalloc(&a1.next_chunk, 8));
    A2 a2{}; a2.stream = stream_b; a2.n_nodes = n_nodes; a2.disp = disp; a2.n_buckets = w->buckets; a2.recs = recs; a2.n_recs = w->rec_slots;
    a2.counts = counts; a2.n_counters = w->count_slots * 2; a2.hit_frac = (uint32_t)(w->hits / w->nodes * 1024.0 + 0.5); a2.sink = sink;
    if (hits_path) {
        FILE *f = fopen(hits_path, "rb");
        if (!f) { perror(hits_path); return 1; }
        fseek(f, 0, SEEK_END); const long nb = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint32_t> h((size_t)nb / 4);
        if (fread(h.data(), 4, h.size(), f) != h.size()) { fprintf(stderr, "short read\n"); return 1; }
        fclose(f);
        uint32_t *d; CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        a2.hits = d; a2.n_hits = h.size();
    }
    const uint32_t grid2 = (uint32_t)(n_cu * w2 / 4);                      // blocks of 256 threads = four waves
    CHECK(hipFuncSetAttribute((const void *)k1_text, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s1, s2; CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t b1, e1, b2, e2; CHECK(hipEventCreate(&b1)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&b2)); CHECK(hipEventCreate(&e2));
    printf("split %s: %.3f GB of text, %llu stripes x %u nodes = %.1f M nodes (%.0f MB of node stream); K1 %d workers per CU, K2 %d waves per CU; %s count updates\n",
           w->name, w->text / 1e9, (unsigned long long)n_stripes, nps, n_nodes / 1e6, n_nodes * 8 / 1e6, w1, w2, hits_path ? "real" : "uniform");
    const uint64_t n_aln = (uint64_t)(w->aln);
    A3 a3{}; a3.stream = stream_b; a3.n_aln = n_aln < n_nodes ? n_aln : n_nodes; a3.disp = disp; a3.n_buckets = w->buckets; a3.recs = recs; a3.n_recs = w->rec_slots;
    a3.counts = counts; a3.n_counters = w->count_slots * 2; a3.hits = a2.hits; a3.n_hits = a2.n_hits; a3.hit_frac = a2.hit_frac; a3.sink = sink;
    a3.k_mean16 = (uint32_t)((w->nodes / w->aln - 1.0) * 2.0 / 15.0 * 128.0 + 0.5);
    for (int what = 0; what < 6; ++what) {
        const bool run1 = what == 0 || what == 2 || what == 3 || what == 5, run2 = what == 1 || what == 2 || what == 3, run3 = what == 4 || what == 5;
        a2.mode = what == 3 ? 1u : 0u;
        float best = 1e9f, sum = 0, s1sum = 0, s2sum = 0;
        for (int i = 0; i < reps + 2; ++i) {
            CHECK(hipMemsetAsync(a1.next_chunk, 0, 8, s1));
            CHECK(hipDeviceSynchronize());
            if (run1) { CHECK(hipEventRecord(b1, s1)); hipLaunchKernelGGL(k1_text, dim3(grid1), dim3(WG), LDS_WORKER, s1, a1); CHECK(hipEventRecord(e1, s1)); }
            if (run2) { CHECK(hipEventRecord(b2, s2)); hipLaunchKernelGGL(k2_tables, dim3(grid2), dim3(256), 0, s2, a2); CHECK(hipEventRecord(e2, s2)); }
            if (run3) { CHECK(hipEventRecord(b2, s2)); hipLaunchKernelGGL(k3_follow, dim3(grid2), dim3(256), 0, s2, a3); CHECK(hipEventRecord(e2, s2)); }
            CHECK(hipDeviceSynchronize());
            float m1 = 0, m2 = 0, span = 0;
            if (run1) CHECK(hipEventElapsedTime(&m1, b1, e1));
            if (run2 || run3) CHECK(hipEventElapsedTime(&m2, b2, e2));
            if (run1 && (run2 || run3)) { float x; CHECK(hipEventElapsedTime(&x, b1, e2)); float y; CHECK(hipEventElapsedTime(&y, b1, e1)); span = x > y ? x : y; }
            else span = run1 ? m1 : m2;
            if (i >= 2) { sum += span; best = span < best ? span : best; s1sum += m1; s2sum += m2; }
        }
        static const char *names[] = {"K1 alone (text -> node stream)", "K2 alone (node stream -> displacement -> record -> count update)", "both at once",
                                      "both at once, K2 with a 32-byte record (two loads)",
                                      "K3 alone (one alignment per lane: first node by name, the rest through the record before)", "K1 and K3 at once"};
        printf("  %-68s  span mean %.4f ms  best %.4f ms   (K1 %.4f, K2 %.4f)\n", names[what], sum / reps, best, s1sum / reps, s2sum / reps);
    }
    return 0;
}
