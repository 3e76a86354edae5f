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
// r05: the one division of the work without an atomic per hit — mode bit 6 — measured on its memory AND LDS side:
//   a worker stages its hits by slot range in LDS (bins of 32 768 counters: 7 at configs[2], 33 at the configs[3] shard; two halves of 32
//   16-bit entries per bin, one LDS atomic per hit for its place), a lane that fills a half writes its 64 bytes to the worker's own
//   segment of that bin's log (no global atomic, one full 64-byte line per 32 hits); k_histogram then counts every bin in LDS
//   (128 KB of counters per workgroup, LDS atomics) and adds them to the count vector, coalesced.  The LDS the staging takes is
//   deducted from the workers per CU (11 520 B + 132 B a bin, in the hardware's 1 280-byte granules).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/skeleton tools/ubench/skeleton.hip ; tools/ubench/skeleton c3|c4shard [reps] [hits.u32] [workers per CU] [log]
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
    // mode bit 6: hits -> binned logs instead of atomics
    uint32_t n_bins;           // bins of BIN counters
    uint16_t *log;             // [bin][worker][seg_cap] 16-bit entries (counter index inside its bin)
    uint32_t *log_fill;        // [bin][worker] entries written
    uint32_t seg_cap;
};
constexpr uint32_t BIN_SHIFT = 15, BIN = 1u << BIN_SHIFT, STAGE_HALF = 32;

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
    // staging of the hits by bin (mode bit 6), behind the main kernel's 11 520 bytes: per bin a fill counter, a write cursor and 2 x 32 entries
    uint32_t *fill = (uint32_t *)(lds + LDS_BYTES_14), *wr = fill + a.n_bins;
    uint16_t *stage = (uint16_t *)(wr + a.n_bins);
    if (a.mode & 64u) { for (uint32_t b = lane; b < a.n_bins; b += WG) { fill[b] = 0; wr[b] = 0; } __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    auto flush_half = [&](uint32_t b, uint32_t half, uint32_t n_valid) {        // 64 bytes of bin b's staged entries -> the worker's segment of the bin's log
        const uint4 *src = (const uint4 *)(stage + (size_t)b * 2 * STAGE_HALF + half * STAGE_HALF);
        const uint32_t w = wr[b];
        if (w + STAGE_HALF <= a.seg_cap) {
            uint4 *dst = (uint4 *)(a.log + ((size_t)b * gridDim.x + blockIdx.x) * a.seg_cap + w);
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
        }
        wr[b] = w + n_valid;
    };
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
                if (a.mode & 64u) {
                    // one LDS atomic per hit for its place among its bin's staged entries; the lane that fills a half sends it off
                    const bool on = lane < a.a_act;
                    const uint32_t hx = hv + (acc & a.zero), b = hx >> BIN_SHIFT;
                    uint32_t pos = 0;
                    if (on) { pos = atomicAdd(&fill[b], 1u); stage[(size_t)b * 2 * STAGE_HALF + (pos & (2 * STAGE_HALF - 1))] = (uint16_t)(hx & (BIN - 1)); }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (on && (pos & (STAGE_HALF - 1)) == STAGE_HALF - 1) flush_half(b, (pos / STAGE_HALF) & 1u, STAGE_HALF);
                } else
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
    if (a.mode & 64u) {                                                  // what is left in the stage: partial halves (padded), then the segments' fills
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t b = lane; b < a.n_bins; b += WG) {
            const uint32_t f = fill[b], rest = f & (STAGE_HALF - 1);
            if (rest) flush_half(b, (f / STAGE_HALF) & 1u, rest);
            a.log_fill[(size_t)b * gridDim.x + blockIdx.x] = wr[b] < a.seg_cap ? wr[b] : a.seg_cap;
        }
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

// bin = blockIdx.y; the workers' segments of the bin are shared out over gridDim.x workgroups; every workgroup counts its segments in LDS
// (BIN counters) and adds what it has to the count vector (consecutive lanes, consecutive counters)
__global__ __launch_bounds__(256) void k_histogram(const uint16_t *log, const uint32_t *log_fill, uint32_t n_workers, uint32_t seg_cap, unsigned int *counts, uint32_t n_counters) {
    extern __shared__ uint32_t hist[];
    const uint32_t b = blockIdx.y;
    for (uint32_t i = threadIdx.x; i < BIN; i += 256) hist[i] = 0;
    __syncthreads();
    for (uint32_t w = blockIdx.x; w < n_workers; w += gridDim.x) {
        const uint32_t n = log_fill[(size_t)b * n_workers + w];
        const uint16_t *seg = log + ((size_t)b * n_workers + w) * seg_cap;
        for (uint32_t i = threadIdx.x * 8; i < n; i += 256 * 8) {         // 16 bytes per lane
            const uint4 v = *(const uint4 *)(seg + i);
            const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) if (i + k < n) atomicAdd(&hist[(e[k >> 1] >> (16 * (k & 1))) & 0xFFFFu], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < BIN; i += 256) {
        const uint32_t c = hist[i], at = b * BIN + i;
        if (c && at < n_counters) atomicAdd(&counts[at], c);
    }
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
    const bool with_log = argc > 5 && !strcmp(argv[5], "log");
    for (const auto &r : runs) {
        if (with_log) break;
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
    if (with_log) {
        // ---- the division without an atomic per hit: staged by bin in LDS -> per-worker log segments -> k_histogram ----
        a.n_bins = (a.n_counters + BIN - 1) / BIN;
        const uint32_t lds_bytes = (LDS_BYTES_14 + a.n_bins * (8 + 2 * STAGE_HALF * 2) + 1279) / 1280 * 1280;
        const int fit = (int)(160 * 1024 / lds_bytes), pc = fit < per_cu ? fit : per_cu;
        const uint32_t g2 = (uint32_t)(n_cu * pc);
        a.region = (uint64_t)((double)((w->text + g2 - 1) / g2) * 0.85) / TEXT * TEXT;
        const double per_seg = w->hits / (double)a.n_bins / (double)g2;
        a.seg_cap = ((uint32_t)(per_seg * 2.0) + 4 * STAGE_HALF + 63) / 64 * 64;     // (uniform bins: twice the mean and a little; a full segment drops what does not fit and says so)
        const size_t log_entries = (size_t)a.n_bins * g2 * a.seg_cap;
        CHECK(hipMalloc(&a.log, log_entries * 2)); CHECK(hipMalloc(&a.log_fill, (size_t)a.n_bins * g2 * 4));
        printf("binned logs: %u bins of %u counters, %u B of LDS per worker -> %d workers per CU (%u in all), segments of %u entries (%.1f MB of log)\n",
               a.n_bins, BIN, lds_bytes, pc, g2, a.seg_cap, log_entries * 2 / 1e6);
        CHECK(hipFuncSetAttribute((const void *)k_histogram, hipFuncAttributeMaxDynamicSharedMemorySize, BIN * 4));
        const struct { const char *what; uint32_t mode; } lruns[] = {
            {"text + tables + count updates, at this many workers (the shipped division, for comparison)", 0u},
            {"text + tables, no counting at all, at this many workers", 4u},
            {"text + tables + hits staged by bin in LDS and logged", 64u},
            {"hits staged and logged only (no text, no tables)", 64u | 3u}};
        for (const auto &r : lruns) {
            a.mode = r.mode;
            float sum = 0, best = 1e9f, hsum = 0, hbest = 1e9f;
            for (int i = 0; i < reps + 2; ++i) {
                CHECK(hipMemsetAsync(a.next_chunk, 0, 8));
                CHECK(hipMemsetAsync(counts, 0, (size_t)w->count_slots * 8));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_skeleton, dim3(g2), dim3(WG), lds_bytes, 0, a);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                float hms = 0;
                if (r.mode & 64u) {
                    const uint32_t split = 256 / a.n_bins + 1;                       // about one workgroup per CU
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_histogram, dim3(split, a.n_bins), dim3(256), BIN * 4, 0, a.log, a.log_fill, g2, a.seg_cap, counts, a.n_counters);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&hms, e0, e1));
                }
                if (i >= 2) { sum += ms; best = ms < best ? ms : best; hsum += hms; hbest = hms < hbest ? hms : hbest; }
            }
            printf("  %-92s  mean %.4f ms  best %.4f ms", r.what, sum / reps, best);
            if (r.mode & 64u) {
                std::vector<unsigned int> hc((size_t)w->count_slots * 2);
                CHECK(hipMemcpy(hc.data(), counts, hc.size() * 4, hipMemcpyDeviceToHost));
                unsigned long long tot = 0; for (unsigned int v : hc) tot += v;
                std::vector<uint32_t> lf((size_t)a.n_bins * g2);
                CHECK(hipMemcpy(lf.data(), a.log_fill, lf.size() * 4, hipMemcpyDeviceToHost));
                unsigned long long logged = 0, full = 0; for (uint32_t v : lf) { logged += v; full += v >= a.seg_cap; }
                printf("  + k_histogram mean %.4f best %.4f ms = %.4f ms  (%llu hits logged, %llu counted, %llu full segments)", hsum / reps, hbest, sum / reps + hsum / reps, logged, tot, full);
            }
            printf("\n");
        }
    }
    return 0;
}
