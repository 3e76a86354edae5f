// measurement only — the memory-side skeleton of k_classify_main (profiles/r04/speed_of_light.txt): a kernel that issues the main
// kernel's GLOBAL memory operations and nothing else, in the main kernel's geometry (one 64-lane wave per workgroup, fourteen
// workers per CU through 11 520 B of dynamic LDS each, every worker a contiguous share of the text walked in 8 KB stripes):
//   per stripe   8 x global_load_dwordx4 nt per lane (the text, once, 16 B per lane, coalesced), first half before the second;
//                the text goes through LDS (ds_write_b128 / one ds_read) so that the loads cannot be dropped
//   per pass     (3 or 4 per stripe, 860 k per launch at configs[2]) on n_act lanes:
//                  one 2-byte load from the displacement array at a random bucket,
//                  then — its address depends on the loaded value — four 16-byte loads of one random 64-byte record,
//                  then on a_act lanes one 32-bit no-return atomic add into the count vector at a random counter, pairs of
//                  neighbouring lanes sharing a counter as often as the real hits do (35.5 M lanes -> 20 M transactions)
// Addresses are uniform over the tables (the synthetic reads start at uniform positions: so are the real ones) and come from a
// counter-based hash of (worker, stripe, pass, lane): a dozen integer instructions per pass, no byte work, no lists, no LDS
// round trips beyond the staging.  The table sizes and per-launch operation counts are the main kernel's (bench.py config block and
// profiles/r04: SQ / TCC counters).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/skeleton tools/ubench/skeleton.hip ; tools/ubench/skeleton c3|c4shard [reps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t WG = 64, TEXT = 8192, HALF = 4096, LDS_BYTES_14 = 11520;

struct Args {
    const uint8_t *text; uint64_t n_bytes; uint64_t region;
    const uint16_t *disp; uint32_t n_buckets;
    const uint4 *recs; uint32_t n_recs;
    unsigned int *counts; uint32_t n_counters;
    uint32_t pass_frac;        // of 1024: stripes with a fourth pass
    uint32_t n_act, a_act;     // lanes of a pass that look a node up / that count a hit
    uint32_t pair_frac;        // of 1024: a counting lane takes the counter of the lane below
    uint32_t mode;             // bit 0: no text, bit 1: no table loads, bit 2: no atomics, bit 3: two passes travel together,
                               // bit 4: one 16-byte load per record instead of four, bit 5: four lanes share a record's line (quad layout)
    uint64_t small;            // the text behind grid * region goes in chunks of this many bytes to whoever is free next (0: even shares)
    unsigned long long *next_chunk;
    uint32_t zero;             // 0 (keeps the dependent addresses dependent)
    const uint32_t *hits; uint64_t n_hits;   // the real kernel's count updates in file order (tools/sol_hits.py), or null: uniform counters
    unsigned long long *sink;
};

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
__device__ inline uint32_t mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

__global__ __launch_bounds__(WG, 4) void k_skeleton(Args a) {
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
    uint64_t hit_at = a.n_hits ? (uint64_t)blockIdx.x * (a.n_hits / gridDim.x) : 0;
  uint32_t s = 0;
  for (;;) {                                                             // chunks: a fixed first one, then small ones from one counter (as the main kernel)
    if (!(a.mode & 1u)) fetch_half(pos);
    for (; pos < end; pos += TEXT, ++s) {
        if (!(a.mode & 1u)) {
#pragma unroll
            for (uint32_t h = 0; h < 2; ++h) {
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) *(uint4 *)(lds + h * HALF + (i * WG + lane) * 16) = pf[i];
                if (h == 0) fetch_half(pos + HALF);
                else if (pos + TEXT < end) fetch_half(pos + TEXT);         // the next stripe's first half travels during the passes
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                acc ^= *(const uint32_t *)(lds + h * HALF + ((lane * 68u) & (HALF - 4u)));
            }
        }
        const uint32_t key = mix(blockIdx.x * 0x9E3779B1u + s);
        const uint32_t n_pass = 3u + ((key & 1023u) < a.pass_frac ? 1u : 0u);
        for (uint32_t p = 0; p < n_pass; ++p) {
            const uint32_t r = mix(key + p * 0x85EBCA77u + lane * 0xC2B2AE3Du);
            if (!(a.mode & 2u)) {
                if (a.mode & 8u) {
                    // measurement variant: two passes travel together (both displacement loads, then both passes' record loads in flight)
                    const bool two = p + 1 < n_pass;
                    const uint32_t rb = mix(key + (p + 1) * 0x85EBCA77u + lane * 0xC2B2AE3Du);
                    uint32_t d0 = 0, d1 = 0;
                    if (lane < a.n_act) { d0 = a.disp[mulhi(r, a.n_buckets)]; if (two) d1 = a.disp[mulhi(rb, a.n_buckets)]; }
                    if (lane < a.n_act) {
                        const uint4 *e0 = a.recs + (size_t)(mulhi(mix(r ^ 0x5bd1e995u), a.n_recs) + (d0 & a.zero)) * 4;
                        const uint4 *e1 = a.recs + (size_t)(mulhi(mix(rb ^ 0x5bd1e995u), a.n_recs) + (d1 & a.zero)) * 4;
                        const uint4 x0 = e0[0], x1 = e0[1], x2 = e0[2], x3 = e0[3];
                        acc ^= x0.x ^ x0.w ^ x1.z ^ x2.x ^ x2.z ^ x3.x ^ x3.z;
                        if (two) { const uint4 y0 = e1[0], y1 = e1[1], y2 = e1[2], y3 = e1[3]; acc ^= y0.x ^ y0.w ^ y1.z ^ y2.x ^ y2.z ^ y3.x ^ y3.z; }
                    }
                } else if (a.mode & 32u) {
                    // measurement variant: the four lanes of a quad read ONE record's 64 bytes per instruction (16 lines per instruction
                    // instead of 64; what it would cost to hand every lane its own record's words is not in here)
                    uint32_t d = 0;
                    if (lane < a.n_act) d = a.disp[mulhi(r, a.n_buckets)];
                    const uint32_t slot = mulhi(mix(r ^ 0x5bd1e995u), a.n_recs) + (d & a.zero);
                    uint4 q[4];
#define QLOAD(i) { const uint32_t si = (uint32_t)__builtin_amdgcn_mov_dpp((int)slot, (i) * 0x55, 0xF, 0xF, false); q[i] = a.recs[(size_t)si * 4 + (lane & 3u)]; }   /* quad_perm: [i, i, i, i] */
                    QLOAD(0) QLOAD(1) QLOAD(2) QLOAD(3)
#undef QLOAD
                    acc ^= q[0].x ^ q[1].y ^ q[2].z ^ q[3].w ^ q[0].w ^ q[1].x ^ q[2].y;
                } else if (lane < a.n_act) {
                    const uint32_t d = a.disp[mulhi(r, a.n_buckets)];
                    const uint32_t slot = mulhi(mix(r ^ 0x5bd1e995u), a.n_recs) + (d & a.zero);   // (depends on the loaded value; a.zero = 0, which the compiler does not know)
                    const uint4 *e = a.recs + (size_t)slot * 4;
                    if (a.mode & 16u) { const uint4 r0 = e[0]; acc ^= r0.x ^ r0.w; }
                    else {
                        const uint4 r0 = e[0], r1 = e[1], r2 = e[2], r3 = e[3];
                        acc ^= r0.x ^ r0.w ^ r1.z ^ r2.x ^ r2.z ^ r3.x ^ r3.z;
                    }
                }
            }
            for (uint32_t q = 0; q < ((a.mode & 8u) && p + 1 < n_pass ? 2u : 1u); ++q) {
                if (a.mode & 4u) break;
                uint32_t hv;
                if (a.hits) {                                            // the real updates, a_act consecutive ones per pass (a worker walks its own stretch of them)
                    hv = a.hits[(hit_at + lane) % a.n_hits];
                    hit_at += a.a_act;
                } else {
                    hv = mulhi(mix(r + 0x27D4EB2Fu + q), a.n_counters);
                    const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp((int)hv, (int)hv, 0x138, 0xF, 0xF, false);   // wave_shr:1
                    if ((mix(r ^ 0x165667B1u) & 1023u) < a.pair_frac) hv = below;
                }
                // (the real kernel counts a hit only after the record has arrived: so does this)
                if (lane < a.a_act) atomicAdd(&a.counts[hv + (acc & a.zero)], 1u);
            }
            if ((a.mode & 8u) && p + 1 < n_pass) ++p;
        }
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

struct Workload { const char *name; uint64_t text; uint32_t rec_slots, buckets, count_slots; double passes, nodes, hits, hit_txn; };
// text bytes, record slots (1.25 n_nodes + 16), displacement buckets (n_nodes / 3 + 1), count slots; per launch: node passes, path nodes
// (one displacement + one record each), counted hits and the transactions they make after the TA has merged equal addresses
static const Workload WL[] = {
    {"c3", 2133165175ull, 247631, 66031, 104881, 0.860e6, 46.63e6, 35.53e6, 19.98e6},
    {"c4shard", 2693024596ull, 1237826, 330083, 524826, 1.074e6, 58.22e6, 44.34e6, 25.02e6},
};

int main(int argc, char **argv) {
    const char *name = argc > 1 ? argv[1] : "c3";
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const char *hits_path = argc > 3 ? argv[3] : nullptr;     // u32 per counted hit (counter index = slot * 2 + allele), file order
    const int per_cu_arg = argc > 4 ? atoi(argv[4]) : 14;
    const Workload *w = nullptr;
    for (const Workload &x : WL) if (!strcmp(x.name, name)) w = &x;
    if (!w) { fprintf(stderr, "workload c3 | c4shard\n"); return 1; }
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount, per_cu = per_cu_arg;
    const uint64_t n_stripes = (w->text + TEXT - 1) / TEXT;
    Args a{};
    const uint64_t pad = TEXT * 2 + 64;
    uint8_t *text; CHECK(hipMalloc(&text, w->text + pad)); CHECK(hipMemset(text, 0x41, w->text + pad));
    uint16_t *disp; CHECK(hipMalloc(&disp, (size_t)w->buckets * 2)); CHECK(hipMemset(disp, 0, (size_t)w->buckets * 2));
    uint4 *recs; CHECK(hipMalloc(&recs, (size_t)w->rec_slots * 64)); CHECK(hipMemset(recs, 0x5A, (size_t)w->rec_slots * 64));
    unsigned int *counts; CHECK(hipMalloc(&counts, (size_t)w->count_slots * 8)); CHECK(hipMemset(counts, 0, (size_t)w->count_slots * 8));
    unsigned long long *sink; CHECK(hipMalloc(&sink, 8));
    const uint32_t grid = (uint32_t)(n_cu * per_cu);
    a.text = text; a.n_bytes = w->text;
    // a worker's first chunk is 85 % of an even share, the rest goes in 64 KB chunks to whoever is free next (svjg_capi.hip: main_launch_setup)
    a.region = (uint64_t)((double)((w->text + grid - 1) / grid) * 0.85) / TEXT * TEXT;
    a.small = 65536;
    CHECK(hipMalloc(&a.next_chunk, 8));
    a.disp = disp; a.n_buckets = w->buckets; a.recs = recs; a.n_recs = w->rec_slots; a.counts = counts; a.n_counters = w->count_slots * 2;
    const double ppstripe = w->passes / (double)n_stripes;
    a.pass_frac = (uint32_t)((ppstripe - 3.0) * 1024.0 + 0.5);
    a.n_act = (uint32_t)(w->nodes / w->passes + 0.5);
    a.a_act = (uint32_t)(w->hits / w->passes + 0.5);
    a.pair_frac = (uint32_t)((1.0 - w->hit_txn / w->hits) * 1024.0 + 0.5);
    a.sink = sink;
    if (hits_path) {
        FILE *f = fopen(hits_path, "rb");
        if (!f) { perror(hits_path); return 1; }
        fseek(f, 0, SEEK_END); const long nb = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint32_t> h((size_t)nb / 4);
        if (fread(h.data(), 4, h.size(), f) != h.size()) { fprintf(stderr, "short read\n"); return 1; }
        fclose(f);
        uint32_t *d; CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        a.hits = d; a.n_hits = h.size();
        printf("count updates: %zu real ones from %s\n", h.size(), hits_path);
    }
    CHECK(hipFuncSetAttribute((const void *)k_skeleton, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("skeleton %s: %.3f GB of text in %llu stripes, %u workers, %.2f passes per stripe, %u lookup lanes and %u counting lanes per pass, %.1f %% of them on the counter of the lane below\n",
           w->name, w->text / 1e9, (unsigned long long)n_stripes, grid, 3.0 + a.pass_frac / 1024.0, a.n_act, a.a_act, a.pair_frac / 10.24);
    const struct { const char *what; uint32_t mode; } runs[] = {
        {"text + tables + count updates (everything the main kernel asks of the memory)", 0u},
        {"text only", 6u}, {"tables only (displacement -> record)", 5u}, {"count updates only", 3u},
        {"text + tables", 4u}, {"text + count updates", 2u},
        {"everything, two passes travelling together (both displacement loads, then both passes' records in flight)", 8u},
        {"tables only, two passes travelling together", 13u},
        {"tables only, one 16-byte load per record instead of four (same lines asked of the L2)", 21u},
        {"tables only, the four lanes of a quad read one record per instruction (16 lines per instruction instead of 64)", 37u},
        {"everything, quad layout of the record loads", 32u}};
    for (const auto &r : runs) {
        a.mode = r.mode;
        float best = 1e9f, sum = 0;
        for (int i = 0; i < reps + 2; ++i) {
            CHECK(hipMemsetAsync(a.next_chunk, 0, 8));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_skeleton, dim3(grid), dim3(WG), per_cu >= 16 ? 10240 : LDS_BYTES_14, 0, a);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 2) { sum += ms; best = ms < best ? ms : best; }
        }
        printf("  %-82s  mean %.4f ms  best %.4f ms\n", r.what, sum / reps, best);
    }
    return 0;
}
