// libsvjg_hip.so — host side of the C ABI declared in include/svjg.h.
// One context = one MI355X, one HIP stream; kernels live in svjg_kernels.h.
#include "svjg_kernels.h"
#include "svjg_host_tables.h"
#include "svjg_pass.h"
#include <rccl/rccl.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <string>
#include <thread>
#include <vector>

using namespace svjg;
typedef struct svjg_ctx svjg_ctx;

static thread_local std::string g_init_error;

// The main kernel's tables (perfect hash of the node names, link table) are built on the host from the graph: 0.25 s at 100 k SVs,
// 2.5 s at 500 k.  A process that gives several contexts the same graph (one per GPU: filter-alignments.py, bench.py --gpus N)
// builds them once: the most recent build is kept, keyed by the sizes and two digests of everything the build reads (graph_digest); the contexts' loads may come from several threads at once (the first builds, the others wait for it).
#include <memory>
#include <mutex>
namespace {
struct TablesCache {
    std::mutex mu;
    uint64_t key[6] = {0, 0, 0, 0, 0, 0};
    std::shared_ptr<const KernelTables> kt;
} g_tables;

uint64_t fold64(const void *p, size_t n, uint64_t h) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    if (n & 7) { uint64_t w = 0; memcpy(&w, b + (n & ~(size_t)7), n & 7); h = (h ^ w ^ ((uint64_t)(n & 7) << 56)) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }   // (the last n % 8 bytes count too)
    return h;
}

// everything build_kernel_tables reads — node, edge and hit arrays, the chromosome names' BYTES (every node name is spelled from
// them), chrom_off, chrom_node_lo, d_over, flags — under two seeds.  No addresses: a freed graph's arrays are handed out again.
uint64_t graph_digest(const svjg_graph &g, uint64_t seed) {
    uint64_t h = seed;
    h = fold64(g.nodes, (size_t)(g.n_nodes + 1) * sizeof(svjg_node), h);
    h = fold64(g.edges, (size_t)g.n_edges * sizeof(svjg_edge), h ^ g.n_edges);
    h = fold64(g.hits, (size_t)g.n_hits * sizeof(uint32_t), h ^ g.n_hits);
    h = fold64(g.chrom_off, (size_t)(g.n_chrom + 1) * sizeof(uint32_t), h ^ g.n_chrom);
    h = fold64(g.chrom_node_lo, (size_t)(g.n_chrom + 1) * sizeof(uint32_t), h);
    h = fold64(g.chrom_names, (size_t)g.chrom_off[g.n_chrom], h);
    const uint64_t tail[3] = {g.d_over, g.flags, g.n_slots};
    return fold64(tail, sizeof tail, h);
}

std::shared_ptr<const KernelTables> kernel_tables_for(const svjg_graph &g) {
    const uint64_t key[6] = {g.n_nodes, g.n_edges, g.n_hits, g.n_chrom, graph_digest(g, 1), graph_digest(g, 0xD6E8FEB86659FD93ull)};
    std::lock_guard<std::mutex> lk(g_tables.mu);              // (held while building: a second loader of the same graph waits, then reuses)
    if (!g_tables.kt || memcmp(key, g_tables.key, sizeof key) != 0) {
        g_tables.kt = std::make_shared<const KernelTables>(build_kernel_tables(g));
        memcpy(g_tables.key, key, sizeof key);
    }
    return g_tables.kt;
}
}  // namespace

extern "C" void svjg_release_host_tables(void) {
    std::lock_guard<std::mutex> lk(g_tables.mu);
    g_tables.kt.reset();
    memset(g_tables.key, 0, sizeof g_tables.key);
}

// ingest geometry: a file range goes to HBM through pinned buffers filled by STAGE_THREADS host threads (page cache ->
// pinned piece by pread -> asynchronous copy on the thread's stream; the next piece is read while the previous one is on
// the bus)
constexpr int STAGE_THREADS = 4;
constexpr uint64_t STAGE_PIECE = 8ull << 20;

struct svjg_ctx {
    int device = 0;
    int n_cu = 256;
    int occ_main = 0;                    // workgroups of k_classify_main one CU holds
    hipStream_t stream = nullptr, copy_stream = nullptr;
    hipEvent_t ev[6] = {};
    std::string err;
    // graph
    bool have_graph = false, have_counts = false;
    svjg_node *d_nodes = nullptr;  svjg_edge *d_edges = nullptr;  uint32_t *d_hits = nullptr;
    uint8_t *d_cnames = nullptr;   uint32_t *d_coff = nullptr, *d_clo = nullptr, *d_chash = nullptr, *d_names = nullptr, *d_links = nullptr, *d_nok = nullptr;  uint16_t *d_disp = nullptr;  uint32_t *d_ihits = nullptr, *d_pfx = nullptr;
    GraphView gv{};
    uint32_t names_len = 0, gflags = 0, n_slots = 0;
    unsigned long long *d_counts = nullptr, *d_snap = nullptr;
    // text
    uint8_t *d_gaf = nullptr;  uint64_t gaf_cap = 0, gaf_bytes = 0;  bool have_gaf = false;
    // ingest: pinned staging buffers, two per feeder thread, each feeder with a copy stream of its own
    char *h_stage[STAGE_THREADS * 2] = {};
    hipStream_t stage_stream[STAGE_THREADS] = {};
    hipEvent_t stage_ev[STAGE_THREADS * 2] = {};
    // outputs
    uint64_t *d_deferred = nullptr;  uint64_t deferred_cap = 0;
    svjg_hitrec *d_recs = nullptr;   uint64_t rec_cap = 0;
    uint64_t *d_host = nullptr;      uint64_t host_cap = 0;     // offsets of the lines set aside for the host (SVJG_EXC_ASK_HOST)
    DevStatus *d_st = nullptr;
    unsigned long long *d_dbg = nullptr;
    uint32_t *d_long = nullptr;          // LONG_WORDS words per worker of k_classify_main (ClassifyArgs::long_pre)
    DevStatus *h_stp = nullptr;          // host twin of d_st in pinned memory: the status copies are asynchronous in both directions
    DevStatus &hs() { return *h_stp; }
    uint64_t total_deferred = 0;
    // genotype scratch
    dd *d_logfact = nullptr;  uint32_t logfact_n = 0;  dd *d_bsum = nullptr;
    unsigned int *d_maxn = nullptr;
    void *d_rows = nullptr;  uint64_t rows_cap = 0;
    void *h_rows = nullptr;  uint64_t h_rows_cap = 0;   // pinned twin of d_rows
    uint64_t geno_rows = 0;                              // rows of the last svjg_genotype / svjg_genotype_view (svjg_genotype_boundary)
    // resident VCF rows of svjg_set_rows / svjg_run_resident: device block (results, row inputs) and the pinned host block the results land in
    struct RunSlot {
        void *d = nullptr;  uint64_t d_cap = 0;  void *h = nullptr;  uint64_t h_cap = 0;  void *h_dev = nullptr;   // h_dev: the pinned block as the device sees it
        unsigned long long *counts = nullptr;  uint64_t counts_cap = 0;   // the pass's own count vector (+ guard words): the next pass may zero its own while this one is genotyped
        hipEvent_t ev[6] = {};  hipEvent_t computed = nullptr, copied = nullptr;
        uint64_t base_offset = 0;  uint32_t min_support = 0;  double err = 0;  bool had_text = false, slow_end_own = false;
    } run[2];
    int run_head = 0, run_tail = 0, run_inflight = 0;
    int counts_in_slot = -1;                             // >= 0: the newest counts live in that slot's vector, not yet in d_counts (fetch_slot_counts)
    void *d_run_in = nullptr;  uint64_t d_run_in_cap = 0;  uint64_t run_rows = 0;  bool have_rows = false;
    // timing of the last calls
    float ms_main = 0, ms_slow = 0, ms_geno = 0;
    // rccl
    ncclComm_t comm = nullptr;
    bool allreduce_second = false;       // svjg_comm_set_stream
    uint64_t part_total = 0, part_need = 0, part_next = 0;   // svjg_gaf_upload_part
};

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return SVJG_E_HIP;                                                                    \
        }                                                                                         \
    } while (0)

extern "C" int svjg_abi_version(void) { return SVJG_ABI_VERSION; }

extern "C" int svjg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *svjg_last_error(const svjg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

extern "C" int svjg_init(int device, svjg_ctx **out) {
    if (!out) return SVJG_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_init_error = "no HIP device visible (libsvjg_hip has no CPU fallback)";
        return SVJG_E_NO_DEVICE;
    }
    if (device < 0 || device >= n) { g_init_error = "device index out of range"; return SVJG_E_ARG; }
    svjg_ctx *c = new svjg_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
        g_init_error = "hipSetDevice / hipStreamCreate failed";
        delete c;
        return SVJG_E_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    for (auto &ev : c->ev) hipEventCreate(&ev);
    if (hipMalloc(&c->d_dbg, 32 * 8) != hipSuccess || hipMalloc(&c->d_st, sizeof(DevStatus)) != hipSuccess || hipMalloc(&c->d_maxn, sizeof(unsigned int)) != hipSuccess ||
        hipHostMalloc((void **)&c->h_stp, 2 * sizeof(DevStatus), hipHostMallocDefault) != hipSuccess) {
        g_init_error = "hipMalloc failed";
        delete c;
        return SVJG_E_NOMEM;
    }
    memset(c->h_stp, 0, 2 * sizeof(DevStatus));
    c->h_stp[1].err = ~0ull;                                  // [1]: a fresh status, never written again (reset_status copies from it without a sync)
    hipFuncSetAttribute((const void *)k_classify_main, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *out = c;
    return 0;
}

static void free_graph(svjg_ctx *c) {
    hipFree(c->d_nodes); hipFree(c->d_edges); hipFree(c->d_hits); hipFree(c->d_cnames); hipFree(c->d_coff);
    hipFree(c->d_clo); hipFree(c->d_chash); hipFree(c->d_counts); hipFree(c->d_snap); hipFree(c->d_names); hipFree(c->d_links); hipFree(c->d_disp); hipFree(c->d_ihits); hipFree(c->d_nok); hipFree(c->d_pfx);
    c->d_pfx = nullptr; c->d_nok = nullptr; c->d_names = nullptr; c->d_links = nullptr; c->d_disp = nullptr; c->d_ihits = nullptr;
    c->d_nodes = nullptr; c->d_edges = nullptr; c->d_hits = nullptr; c->d_cnames = nullptr; c->d_coff = nullptr;
    c->d_clo = nullptr; c->d_chash = nullptr; c->d_counts = nullptr; c->d_snap = nullptr;
    c->have_graph = false; c->have_counts = false;
}

extern "C" void svjg_destroy(svjg_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    if (c->comm) ncclCommDestroy(c->comm);
    free_graph(c);
    hipFree(c->d_long);
    hipFree(c->d_gaf); hipFree(c->d_deferred); hipFree(c->d_recs); hipFree(c->d_host); hipFree(c->d_st); hipFree(c->d_logfact);
    hipFree(c->d_bsum); hipFree(c->d_maxn); hipFree(c->d_rows); hipFree(c->d_run_in);
    for (auto &r : c->run) {
        hipFree(r.d); hipFree(r.counts);
        if (r.h) hipHostFree(r.h);
        for (auto &e : r.ev) if (e) hipEventDestroy(e);
        if (r.computed) hipEventDestroy(r.computed);
        if (r.copied) hipEventDestroy(r.copied);
    }
    if (c->h_rows) hipHostFree(c->h_rows);
    if (c->h_stp) hipHostFree(c->h_stp);
    for (auto &b : c->h_stage) if (b) hipHostFree(b);
    for (auto &ev : c->stage_ev) if (ev) hipEventDestroy(ev);
    for (auto &st : c->stage_stream) if (st) hipStreamDestroy(st);
    for (auto &ev : c->ev) if (ev) hipEventDestroy(ev);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

template <class T>
static int upload(svjg_ctx *c, T **dst, const T *src, uint64_t n, uint64_t extra = 0) {
    HIPCHK(c, hipMalloc((void **)dst, (n + extra + 1) * sizeof(T)));
    if (n) HIPCHK(c, hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return 0;
}

#ifndef SVJG_SMALL_CHUNK
#define SVJG_SMALL_CHUNK 65536
#endif
#ifndef SVJG_FIRST_SHARE
#define SVJG_FIRST_SHARE 0.85
#endif
constexpr uint64_t SMALL_CHUNK = SVJG_SMALL_CHUNK;            // bytes (multiple of 16)
constexpr double FIRST_CHUNK_SHARE = SVJG_FIRST_SHARE;       // of an even share of the text

static int reset_status(svjg_ctx *c, bool all) {
    if (all) {                                                // from the constant copy: the host's own status may be written again at once
        c->hs() = c->h_stp[1]; c->total_deferred = 0;
        HIPCHK(c, hipMemcpyAsync(c->d_st, &c->h_stp[1], sizeof(DevStatus), hipMemcpyHostToDevice, c->stream));
        return 0;
    }
    // (every earlier writer of hs() on the stream — the status read-backs — was followed by a synchronisation)
    c->hs().n_deferred = 0; c->hs().overflow = 0; c->hs().next_chunk = 0;
    HIPCHK(c, hipMemcpyAsync(c->d_st, &c->hs(), sizeof(DevStatus), hipMemcpyHostToDevice, c->stream));
    return 0;
}

extern "C" int svjg_load_graph(svjg_ctx *c, const svjg_graph *g) {
    if (!c || !g || !g->nodes || !g->chrom_off || !g->chrom_node_lo) return SVJG_E_ARG;
    if (g->n_nodes >= 0x7FFFFFFFull || g->n_edges >= 0x7FFFFFFFull || g->n_chrom >= 65535) { c->err = "graph too large"; return SVJG_E_ARG; }
    if (g->d_over >= 0x80000000u) { c->err = "d_over too large"; return SVJG_E_ARG; }   // (the overlap sums of the main kernel are 32 bits wide)
    HIPCHK(c, hipSetDevice(c->device));
    free_graph(c);
    int rc;
    struct Undo { svjg_ctx *c; bool armed = true; ~Undo() { if (armed) { hipStreamSynchronize(c->stream); free_graph(c); } } } undo{c};   // a failure midway leaves nothing behind
    if ((rc = upload(c, &c->d_nodes, g->nodes, g->n_nodes + 1))) return rc;
    if ((rc = upload(c, &c->d_edges, g->edges, g->n_edges))) return rc;
    if ((rc = upload(c, &c->d_hits, g->hits, g->n_hits))) return rc;
    c->names_len = g->chrom_off[g->n_chrom];
    if ((rc = upload(c, &c->d_cnames, (const uint8_t *)g->chrom_names, c->names_len, 8))) return rc;
    if ((rc = upload(c, &c->d_coff, g->chrom_off, g->n_chrom + 1))) return rc;
    if ((rc = upload(c, &c->d_clo, g->chrom_node_lo, g->n_chrom + 1))) return rc;
    std::vector<uint32_t> hash = build_chrom_hash(*g);
    if ((rc = upload(c, &c->d_chash, hash.data(), hash.size()))) return rc;
    const std::shared_ptr<const KernelTables> ktp = kernel_tables_for(*g);   // (built once per graph and process)
    const KernelTables &kt = *ktp;
    if ((rc = upload(c, &c->d_names, kt.names.data(), kt.names.size()))) return rc;
    if ((rc = upload(c, &c->d_links, kt.links.data(), kt.links.size()))) return rc;
    if ((rc = upload(c, &c->d_disp, kt.disp.data(), kt.disp.size()))) return rc;
    if ((rc = upload(c, &c->d_ihits, kt.ihits.data(), kt.ihits.size()))) return rc;
    if ((rc = upload(c, &c->d_nok, kt.node_of_kid.data(), kt.node_of_kid.size()))) return rc;
    if (!kt.name_pfx.empty() && (rc = upload(c, &c->d_pfx, kt.name_pfx.data(), kt.name_pfx.size()))) return rc;   // (names of 49..64 bytes: rare)
    HIPCHK(c, hipStreamSynchronize(c->stream));           // `hash` and `kt` are locals
    c->gv.nodes = c->d_nodes; c->gv.n_nodes = (uint32_t)g->n_nodes; c->gv.edges = c->d_edges; c->gv.hits = c->d_hits;
    c->gv.chrom_names = c->d_cnames; c->gv.chrom_off = c->d_coff; c->gv.chrom_lo = c->d_clo; c->gv.chrom_hash = c->d_chash;
    c->gv.n_chrom = g->n_chrom; c->gv.hash_mask = (uint32_t)hash.size() - 1; c->gv.d_over = g->d_over;
    c->gv.dover_list = (g->flags & SVJG_GRAPH_DOVER_LIST) ? 1u : 0u;
    c->gv.name_pfx = c->d_pfx;
    c->gv.node_of_kid = c->d_nok; c->gv.name_tab = c->d_names; c->gv.name_ihits = c->d_ihits; c->gv.name_disp = c->d_disp; c->gv.name_slots = kt.name_slots; c->gv.name_buckets = kt.name_buckets;
    c->gv.name_complete = (kt.names_left_out == 0 && kt.names_skipped == 0) ? 1u : 0u;
    c->gv.link_tab = c->d_links; c->gv.link_mask = kt.link_mask; c->gv.link_seed = kt.link_seed;
    c->gflags = g->flags;
    if (g->flags & SVJG_GRAPH_DOVER_LIST) c->gflags |= SVJG_GRAPH_ALL_SLOW;   // (the exact routine knows where the reference's TypeError sits)
    if (kt.links_left_out) c->gflags |= SVJG_GRAPH_ALL_SLOW;   // a link the main kernel could not find would be a silent miss
    c->n_slots = g->n_slots;
    HIPCHK(c, hipMalloc((void **)&c->d_counts, ((uint64_t)g->n_slots + GUARD_WORDS) * 8));
    HIPCHK(c, hipMalloc((void **)&c->d_snap, ((uint64_t)g->n_slots + GUARD_WORDS) * 8));
    c->have_graph = true; c->have_counts = true;
    undo.armed = false;
    return svjg_reset_counts(c);
}

extern "C" int svjg_alloc_counts(svjg_ctx *c, uint32_t n_slots) {
    if (!c) return SVJG_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    free_graph(c);
    c->n_slots = n_slots;
    HIPCHK(c, hipMalloc((void **)&c->d_counts, ((uint64_t)n_slots + GUARD_WORDS) * 8));
    HIPCHK(c, hipMalloc((void **)&c->d_snap, ((uint64_t)n_slots + GUARD_WORDS) * 8));
    c->have_counts = true;
    return svjg_reset_counts(c);
}

extern "C" int svjg_reset_counts(svjg_ctx *c) {
    if (!c || !c->have_counts) return SVJG_E_ARG;
    c->counts_in_slot = -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_counts, 0, ((uint64_t)c->n_slots + GUARD_WORDS) * 8, c->stream));
    return reset_status(c, true);                             // (nothing to wait for: the next call on the stream comes behind both)
}

// room for n bytes of text plus the zero padding the kernels read past the end without bounds tests
static int gaf_reserve(svjg_ctx *c, uint64_t n, uint64_t *need_out) {
    const uint64_t need = ((n + 15) & ~15ull) + TEXT + 64;
    if (need > c->gaf_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_gaf); c->d_gaf = nullptr; c->gaf_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_gaf, need));
        c->gaf_cap = need;
    }
    *need_out = need;
    return 0;
}

// bytes [offset, offset + n) of the open file -> d_gaf through the pinned staging buffers
static int staged_upload(svjg_ctx *c, int fd, uint64_t offset, uint64_t n, uint64_t dst_off = 0) {
    for (int i = 0; i < STAGE_THREADS * 2; ++i) {
        if (!c->h_stage[i]) HIPCHK(c, hipHostMalloc((void **)&c->h_stage[i], STAGE_PIECE, hipHostMallocDefault));
        if (!c->stage_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
    }
    for (int t = 0; t < STAGE_THREADS; ++t)
        if (!c->stage_stream[t]) HIPCHK(c, hipStreamCreateWithFlags(&c->stage_stream[t], hipStreamNonBlocking));
    const uint64_t n_pieces = (n + STAGE_PIECE - 1) / STAGE_PIECE;
    std::string errs[STAGE_THREADS];
    int rcs[STAGE_THREADS] = {};
    auto feeder = [&](int t) {
        auto fail = [&](int rc, const std::string &m) { rcs[t] = rc; errs[t] = m; };
        if (hipSetDevice(c->device) != hipSuccess) return fail(SVJG_E_HIP, "hipSetDevice in an ingest thread");
        bool used[2] = {false, false};
        uint64_t k = 0;
        for (uint64_t p = (uint64_t)t; p < n_pieces && !rcs[t]; p += STAGE_THREADS, ++k) {
            const int b = t * 2 + (int)(k & 1);
            const uint64_t a = p * STAGE_PIECE, len = n - a < STAGE_PIECE ? n - a : STAGE_PIECE;
            if (used[k & 1] && hipEventSynchronize(c->stage_ev[b]) != hipSuccess) return fail(SVJG_E_HIP, "hipEventSynchronize (ingest)");
            for (uint64_t got = 0; got < len;) {
                const ssize_t r = pread(fd, c->h_stage[b] + got, len - got, (off_t)(offset + a + got));
                if (r <= 0) return fail(SVJG_E_IO, r == 0 ? "the GAF file is shorter than announced" : std::string("pread: ") + strerror(errno));
                got += (uint64_t)r;
            }
            if (hipMemcpyAsync(c->d_gaf + dst_off + a, c->h_stage[b], len, hipMemcpyHostToDevice, c->stage_stream[t]) != hipSuccess ||
                hipEventRecord(c->stage_ev[b], c->stage_stream[t]) != hipSuccess) return fail(SVJG_E_HIP, "hipMemcpyAsync (ingest)");
            used[k & 1] = true;
        }
        if (hipStreamSynchronize(c->stage_stream[t]) != hipSuccess) fail(SVJG_E_HIP, "hipStreamSynchronize (ingest)");
    };
    std::vector<std::thread> th;
    for (int t = 1; t < STAGE_THREADS; ++t) th.emplace_back(feeder, t);
    feeder(0);
    for (auto &x : th) x.join();
    for (int t = 0; t < STAGE_THREADS; ++t)
        if (rcs[t]) { c->err = errs[t]; return rcs[t]; }
    return 0;
}

// blocks of k_classify_slow a CU holds at a time (LDS and registers decide; asked of the runtime once, 5 on gfx950 today)
static uint32_t lane_blocks_per_cu(svjg_ctx *c) {
    static int per_cu = 0;
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_classify_slow, (int)SLOW_TPB, 0) != hipSuccess || n < 1) n = 4;
        per_cu = n;
    }
    (void)c;
    return (uint32_t)per_cu;
}

static int gaf_finish(svjg_ctx *c, uint64_t n, uint64_t need) {
    HIPCHK(c, hipMemsetAsync(c->d_gaf + n, 0, need - n, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->gaf_bytes = n;
    c->have_gaf = true;
    return 0;
}

extern "C" int svjg_gaf_upload(svjg_ctx *c, const char *gaf, uint64_t n) {
    if (!c || (n && !gaf)) return SVJG_E_ARG;
    if (c->run_inflight) { c->err = "svjg_gaf_upload with a pass in flight (svjg_run_end first: a repeat of that pass would read the new text)"; return SVJG_E_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t need;
    int rc = gaf_reserve(c, n, &need);
    if (rc) return rc;
    c->have_gaf = false;
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_gaf, gaf, n, hipMemcpyHostToDevice, c->stream));   // (the runtime stages pageable memory itself: 47 GB/s measured)
    return gaf_finish(c, n, need);
}

extern "C" int svjg_gaf_upload_part(svjg_ctx *c, const char *gaf, uint64_t n, uint64_t offset, uint64_t capacity, int last) {
    if (!c || (n && !gaf) || n > capacity || offset > capacity - n) return SVJG_E_ARG;
    if (c->run_inflight) { c->err = "svjg_gaf_upload_part with a pass in flight (svjg_run_end first)"; return SVJG_E_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    if (offset == 0) {
        int rc = gaf_reserve(c, capacity, &c->part_need);
        if (rc) return rc;
        c->have_gaf = false; c->part_total = capacity; c->part_next = 0;
    }
    if (capacity != c->part_total || offset != c->part_next || c->have_gaf) { c->err = "svjg_gaf_upload_part: pieces go in ascending order, the first at offset 0"; return SVJG_E_ARG; }
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_gaf + offset, gaf, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));               // (the caller's buffer may go away)
    c->part_next = offset + n;
    if (last) return gaf_finish(c, c->part_next, c->part_need);   // (zero padding from the text's end to the end of the buffer)
    return 0;
}

extern "C" int svjg_gaf_upload_file(svjg_ctx *c, const char *path, uint64_t offset, uint64_t n) {
    if (!c || !path) return SVJG_E_ARG;
    if (c->run_inflight) { c->err = "svjg_gaf_upload_file with a pass in flight (svjg_run_end first)"; return SVJG_E_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t need;
    int rc = gaf_reserve(c, n, &need);
    if (rc) return rc;
    c->have_gaf = false;
    if (n) {
        const int fd = open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) { c->err = std::string("open ") + path + ": " + strerror(errno); return SVJG_E_IO; }
        rc = staged_upload(c, fd, offset, n);
        close(fd);
        if (rc) return rc;
    }
    return gaf_finish(c, n, need);
}

static int ensure(svjg_ctx *c, void **p, uint64_t *cap, uint64_t want, size_t elem, bool keep) {
    if (want <= *cap) return 0;
    // a buffer that keeps its contents grows by at least half: the hit records of a file classified in pieces used to be moved into a
    // buffer a piece larger 2 x 169 times at configs[3] (hipMalloc / copy / hipFree of up to 5.7 GB each: 7 of the 8.9 s of its ingest)
    const uint64_t asked = want;
    if (keep && *cap && want < *cap + *cap / 2) want = *cap + *cap / 2;
    void *q = nullptr;
    if (want != asked && hipMalloc(&q, want * elem) != hipSuccess) { (void)hipGetLastError(); q = nullptr; want = asked; }   // (no room for the margin: what was asked for)
    if (!q) HIPCHK(c, hipMalloc(&q, want * elem));
    if (keep && *p && *cap) HIPCHK(c, hipMemcpyAsync(q, *p, *cap * elem, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(*p);
    *p = q; *cap = want;
    return 0;
}

// arguments and geometry of one k_classify_main launch over the lines of the resident text that lie in [begin, end)
// The passes of svjg_run_begin keep their counts in vectors of their own; the rest of the API works on d_counts: the newest pass's
// vector is copied there when someone asks for it (svjg_get_counts, svjg_genotype, svjg_classify adding to it, ...).
static int fetch_slot_counts(svjg_ctx *c) {
    if (c->counts_in_slot < 0) return 0;
    const int k = c->counts_in_slot;
    c->counts_in_slot = -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->run[k].copied) HIPCHK(c, hipStreamWaitEvent(c->stream, c->run[k].copied, 0));   // (the pass's all-reduce runs on the second stream)
    HIPCHK(c, hipMemcpyAsync(c->d_counts, c->run[k].counts, ((uint64_t)c->n_slots + GUARD_WORDS) * 8, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

static uint64_t deferred_want(const svjg_ctx *c, uint64_t n) { return (c->gflags & SVJG_GRAPH_ALL_SLOW) ? n / 24 + 64 : (n / 4096 + 65536); }

static void main_launch_setup(svjg_ctx *c, uint64_t begin, uint64_t end, uint64_t base_offset, int want_hits, ClassifyArgs &a, uint32_t &grid, size_t &lds) {
    const uint64_t n = end - begin;
    const bool all_slow = (c->gflags & SVJG_GRAPH_ALL_SLOW) != 0;
    a.gaf = c->d_gaf; a.begin = begin; a.n_bytes = end; a.base_offset = base_offset; a.g = c->gv;
    a.all_slow = all_slow; a.want_hits = want_hits != 0;
#ifdef SVJG_ABLATE
    { const char *dg = getenv("SVJG_DIAG"); a.diag = dg ? (uint32_t)atoi(dg) : 0u; }   // ablation knob (measurement builds only)
#endif
    a.counts = c->d_counts; a.deferred = c->d_deferred; a.deferred_cap = c->deferred_cap;
    a.recs = c->d_recs; a.rec_cap = c->rec_cap; a.host_lines = c->d_host; a.host_cap = c->host_cap; a.st = c->d_st; a.dbg = c->d_dbg;
    lds = LDS_MAIN;
    if (c->occ_main < 1) {                                // persistent grid: every CU filled to what LDS / registers admit (asked once)
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_classify_main, (int)WG, lds) != hipSuccess || occ < 1) occ = 1;
        // LDS is handed out in units of LDS_GRANULE bytes (measured: a 15 568-byte workgroup is admitted 9 times per CU where the
        // API says 10); a worker too many would run in a second round behind the others
        const int by_lds = (int)((160u * 1024u) / ((lds + LDS_GRANULE - 1) / LDS_GRANULE * LDS_GRANULE));
        if (by_lds >= 1 && occ > by_lds) occ = by_lds;
        c->occ_main = occ;
        hipFree(c->d_long); c->d_long = nullptr;
#ifdef SVJG_ABLATE
        { const char *oc = getenv("SVJG_OCC"); fprintf(stderr, "[svjg diag] occupancy API: %d workgroups of %u threads per CU (LDS %zu)\n", occ, WG, lds); if (oc && atoi(oc) > 0) c->occ_main = atoi(oc); }
#endif
    }
    // one worker (wave) per resident slot; every worker owns the lines starting in its region of the text.  Small inputs:
    // regions of at least one stripe, fewer workers.
    // A worker's first chunk is most of an even share, fixed; what is left goes in small chunks to whoever is free next
    // (svjg_kernels.h: the workers of a CU do not run equally fast).  Inputs too small for that: even shares of at least a stripe.
    const uint64_t full = (uint64_t)c->n_cu * (uint64_t)c->occ_main;
    if (!c->d_long && hipMalloc((void **)&c->d_long, full * LONG_WORDS * sizeof(uint32_t)) != hipSuccess) c->d_long = nullptr;   // (checked by the callers: a launch without it is refused)
    a.long_pre = c->d_long;
    const uint64_t share = (n + full - 1) / full;
    uint64_t region = (share + 15) & ~15ull;
    a.small = 0;
    if (share >= 8 * SMALL_CHUNK) { region = ((uint64_t)(share * FIRST_CHUNK_SHARE) + 15) & ~15ull; a.small = SMALL_CHUNK; }
    if (region < TEXT) region = TEXT;
    a.region = region;
    const uint64_t want_grid = (n + region - 1) / region;
    grid = (uint32_t)(want_grid < full ? want_grid : full);
}

// the lines of the resident text that lie in [begin, end): begin is a line start, end the byte behind a terminator (or the text's end)
static int classify_range(svjg_ctx *c, uint64_t begin, uint64_t end, uint64_t base_offset, int want_hits) {
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc0 = fetch_slot_counts(c); if (rc0) return rc0; }
    if (end <= begin) return 0;
    const uint64_t n = end - begin;
    uint64_t def_want = deferred_want(c, n);
    uint64_t rec_want = want_hits ? c->hs().n_recs + n / 40 + 65536 : 0;   // (the synthetic workloads: one hit per 60 bytes of text)
    uint64_t host_want = c->hs().n_host + 4096;
    int rc;
    for (int attempt = 0; attempt < 3; ++attempt) {
        if ((rc = ensure(c, (void **)&c->d_deferred, &c->deferred_cap, def_want, sizeof(uint64_t), false))) return rc;
        if ((rc = ensure(c, (void **)&c->d_host, &c->host_cap, host_want, sizeof(uint64_t), true))) return rc;
        if (want_hits && (rc = ensure(c, (void **)&c->d_recs, &c->rec_cap, rec_want, sizeof(svjg_hitrec), true))) return rc;
        // snapshot so that an overflowed attempt can be rolled back
        HIPCHK(c, hipMemcpyAsync(c->d_snap, c->d_counts, ((uint64_t)c->n_slots + GUARD_WORDS) * 8, hipMemcpyDeviceToDevice, c->stream));
        DevStatus before = c->hs();
        if ((rc = reset_status(c, false))) return rc;
        ClassifyArgs a{};
        uint32_t grid = 0;
        size_t lds = 0;
        main_launch_setup(c, begin, end, base_offset, want_hits, a, grid, lds);
        if (!a.long_pre) { c->err = "no memory for the workers' scratch words"; return SVJG_E_NOMEM; }
#ifdef SVJG_TIMING
        if (a.diag & 16u) HIPCHK(c, hipMemsetAsync(c->d_dbg, 0, 32 * 8, c->stream));
#endif
        HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
        hipLaunchKernelGGL(k_classify_main, dim3(grid), dim3(WG), lds, c->stream, a);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        HIPCHK(c, hipMemcpyAsync(&c->hs(), c->d_st, sizeof(DevStatus), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint64_t n_def = c->hs().n_deferred;
#ifdef SVJG_TIMING
        if (a.diag & 16u) {
            unsigned long long d[16];
            HIPCHK(c, hipMemcpy(d, c->d_dbg, sizeof d, hipMemcpyDeviceToHost));
            fprintf(stderr, "[svjg diag] sub-passes of long lines: first sweep counting %llu, first sweep measuring only %llu, second sweep %llu; nodes in sub-passes %llu\n", d[12], d[13], d[14], d[15]);
            fprintf(stderr, "[svjg diag] wave time per phase (sum over waves, counter ticks)  A+B1 %llu  prefix %llu  B2 %llu  R1 %llu  NP: lists %llu  names+disp %llu  records %llu  scan+search %llu  links+atomics %llu  R6+end %llu;  passes %llu (sub-passes of long lines %llu), nodes in them %llu\n",
                    d[0], d[1], d[2], d[3], d[8], d[9], d[4], 0ull, d[6], d[7], d[10], d[11], d[5]);
        }
#endif
        c->ms_slow = 0;
        if (n_def && !(c->hs().overflow & 1u)) {
            HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
            const uint64_t max_blocks = (uint64_t)c->n_cu * 4;              // (53 KB of LDS each: three at a time per CU)
            const uint64_t lane_blocks = (uint64_t)c->n_cu * 9;           // the lane-per-line kernel: more blocks than the CUs hold at a time (5 each): the queue evens out the blocks' different times
            if (n_def <= 30 * max_blocks) {
                // one wave per line: ~52 ns a line with every CU busy (10 k ordinary lines 0.53 ms); the lane-per-line kernel needs ~1.6 ms
                // for any number of lines (sixty-four lines with sixty-four control flows share a wave) and ~4 ns for every further one:
                // they meet at ~31 k lines (profiles/r04/experiments/exact_path.txt)
                hipLaunchKernelGGL(k_classify_slow_wave, dim3((uint32_t)(n_def < max_blocks ? n_def : max_blocks)), dim3(SLOW_TPB), 0, c->stream, a, n_def, 0ull, 0ull);
            } else {
                const uint64_t want_blocks = (n_def + SLOW_TPB - 1) / SLOW_TPB;
                hipLaunchKernelGGL(k_classify_slow, dim3((uint32_t)(want_blocks < lane_blocks ? want_blocks : lane_blocks)), dim3(SLOW_TPB), 0, c->stream, a, n_def, 0ull, 0ull);
            }
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
            HIPCHK(c, hipMemcpyAsync(&c->hs(), c->d_st, sizeof(DevStatus), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipEventElapsedTime(&c->ms_slow, c->ev[2], c->ev[3]));
#ifdef SVJG_TIMING
            if (a.diag & 16u) {
                unsigned long long d[8];
                HIPCHK(c, hipMemcpy(d, c->d_dbg + 16, sizeof d, hipMemcpyDeviceToHost));
                fprintf(stderr, "[svjg diag] one wave per line, the longest any line took per step (counter ticks): terminator + staging %llu  per-line part %llu  piece table %llu  nodes %llu  links %llu\n",
                        d[0], d[1], d[2], d[3], d[4]);
                unsigned long long l[5];
                HIPCHK(c, hipMemcpy(l, c->d_dbg + 24, sizeof l, hipMemcpyDeviceToHost));
                fprintf(stderr, "[svjg diag] one lane per line, ticks summed over the blocks of 64 lines: terminators + staging %llu  per-line part %llu  nodes %llu  links %llu  rest %llu\n",
                        l[0], l[1], l[2], l[3], l[4]);
            }
#endif
        }
        HIPCHK(c, hipEventElapsedTime(&c->ms_main, c->ev[0], c->ev[1]));
        if (!c->hs().overflow) {
            c->total_deferred += n_def;                               // lines that took the exact path
            break;
        }
        // roll back and retry with worst-case buffers
        if (attempt == 2) { c->err = "output buffers overflowed repeatedly"; return SVJG_E_NOMEM; }
        if (c->hs().overflow & 1u) def_want = n / 24 + 64;
        if (c->hs().overflow & 2u) rec_want = before.n_recs + (c->hs().n_recs - before.n_recs) * 2 + n / 24 + 64;
        if (c->hs().overflow & 4u) host_want = before.n_host + n / 24 + 64;
        HIPCHK(c, hipMemcpyAsync(c->d_counts, c->d_snap, ((uint64_t)c->n_slots + GUARD_WORDS) * 8, hipMemcpyDeviceToDevice, c->stream));
        uint64_t keep_err = before.err;
        c->hs() = before; c->hs().err = keep_err;
    }
    if (c->hs().err != ~0ull) return SVJG_E_INPUT;
    return 0;
}

extern "C" int svjg_classify_resident(svjg_ctx *c, uint64_t base_offset, int want_hits) {
    if (!c || !c->have_graph || !c->have_gaf) { if (c) c->err = "classify needs a graph and an uploaded GAF"; return SVJG_E_ARG; }
    return classify_range(c, 0, c->gaf_bytes, base_offset, want_hits);
}

// Upload and classify overlapped: the text goes to HBM in pieces of PIPE_BYTES cut behind line terminators; while the kernels
// work on piece k (this thread, the context's stream), a helper thread copies piece k + 1 (its own streams).  Every piece gets
// PIPE_GAP zero bytes behind it in the device buffer (the kernels read past the end of their text without bounds tests), so
// piece k sits at device offset cut[k] + k * PIPE_GAP and offsets are reported through a per-piece base.
// `fetch(dst_off, src_off, len)` moves bytes [src_off, src_off + len) of the source to d_gaf + dst_off and returns when
// they are there; `last_cut(lo, hi)` = offset behind the last terminator in [lo, hi) of the source, or lo if there is none.
constexpr uint64_t PIPE_BYTES = 128ull << 20;
constexpr uint64_t PIPE_GAP = ((uint64_t)TEXT + 64 + 15) & ~15ull;
template <class Fetch, class Cut>
static int classify_pipelined(svjg_ctx *c, uint64_t n, uint64_t base_offset, int want_hits, Fetch fetch, Cut last_cut) {
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> cuts{0};
    while (cuts.back() < n) {
        const uint64_t lo = cuts.back(), hi = lo + PIPE_BYTES < n ? lo + PIPE_BYTES : n;
        uint64_t cut = hi == n ? n : last_cut(lo, hi);
        if (cut <= lo) cut = hi == n ? n : last_cut(lo, n);               // a line longer than a piece: up to the next terminator behind it
        if (cut <= lo) cut = n;
        cuts.push_back(cut);
    }
    const size_t n_pieces = cuts.size() - 1;
    auto dev_off = [&](size_t k) { return ((cuts[k] + 15) & ~15ull) + k * PIPE_GAP; };   // 16-byte aligned, >= PIPE_GAP - 15 zero bytes in front
    uint64_t need;
    int rc = gaf_reserve(c, dev_off(n_pieces - 1) + (cuts[n_pieces] - cuts[n_pieces - 1]), &need);
    if (rc) return rc;
    c->have_gaf = false;
    for (size_t k = 0; k < n_pieces; ++k) {                                // zero padding behind every piece
        const uint64_t e = dev_off(k) + (cuts[k + 1] - cuts[k]);
        const uint64_t upto = k + 1 < n_pieces ? dev_off(k + 1) : need;
        HIPCHK(c, hipMemsetAsync(c->d_gaf + e, 0, upto - e, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((rc = fetch(dev_off(0), 0, cuts[1]))) return rc;
    for (size_t k = 0; k < n_pieces; ++k) {
        int rc_up = 0;
        std::thread up;
        if (k + 1 < n_pieces) up = std::thread([&, k] { rc_up = fetch(dev_off(k + 1), cuts[k + 1], cuts[k + 2] - cuts[k + 1]); });
        // (reported offsets = base + device offset: the base of a piece takes its place in the buffer out again; unsigned wrap-around is fine)
        rc = classify_range(c, dev_off(k), dev_off(k) + (cuts[k + 1] - cuts[k]), base_offset + cuts[k] - dev_off(k), want_hits);
        if (up.joinable()) up.join();
        if (rc) return rc;
        if (rc_up) return rc_up;
    }
    // (the buffer now holds the pieces with gaps: not a resident text for svjg_classify_resident)
    c->gaf_bytes = 0;
    return 0;
}

extern "C" int svjg_classify(svjg_ctx *c, const char *gaf, uint64_t n, uint64_t base_offset, int want_hits) {
    if (!c || (n && !gaf)) return SVJG_E_ARG;
    if (!c->have_graph) { c->err = "classify needs a graph"; return SVJG_E_ARG; }
    if (n == 0) return svjg_gaf_upload(c, gaf, 0);
    std::string uerr;
    auto fetch = [&](uint64_t dst, uint64_t src, uint64_t len) -> int {
        if (hipSetDevice(c->device) != hipSuccess) { uerr = "hipSetDevice (upload thread)"; return SVJG_E_HIP; }
        if (!c->copy_stream && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { uerr = "hipStreamCreate (upload)"; return SVJG_E_HIP; }
        if (hipMemcpyAsync(c->d_gaf + dst, gaf + src, len, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
            hipStreamSynchronize(c->copy_stream) != hipSuccess) { uerr = "hipMemcpyAsync (upload)"; return SVJG_E_HIP; }
        return 0;
    };
    auto last_cut = [&](uint64_t lo, uint64_t hi) -> uint64_t {
        if (hi == n) {                                                     // forward: the first terminator at or behind lo
            for (uint64_t p = lo; p < n; ++p) if (gaf[p] == '\n') return p + 1;
            return lo;
        }
        const void *q = memrchr(gaf + lo, '\n', hi - lo);
        return q ? (uint64_t)((const char *)q - gaf) + 1 : lo;
    };
    const int rc = classify_pipelined(c, n, base_offset, want_hits, fetch, last_cut);
    if (rc && !uerr.empty()) c->err = uerr;
    return rc;
}

extern "C" int svjg_classify_file(svjg_ctx *c, const char *path, uint64_t offset, uint64_t n, int want_hits) {
    if (!c || !path) return SVJG_E_ARG;
    if (!c->have_graph) { c->err = "classify needs a graph"; return SVJG_E_ARG; }
    if (n == 0) return svjg_gaf_upload(c, nullptr, 0);
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { c->err = std::string("open ") + path + ": " + strerror(errno); return SVJG_E_IO; }
    auto fetch = [&](uint64_t dst, uint64_t src, uint64_t len) -> int { return staged_upload(c, fd, offset + src, len, dst); };
    auto last_cut = [&](uint64_t lo, uint64_t hi) -> uint64_t {
        std::vector<char> buf(1 << 20);
        if (hi == n && lo) {                                               // forward: the first terminator at or behind lo
            for (uint64_t p = lo; p < n;) {
                const ssize_t r = pread(fd, buf.data(), buf.size(), (off_t)(offset + p));
                if (r <= 0) break;
                const void *q = memchr(buf.data(), '\n', (size_t)r);
                if (q) return p + (uint64_t)((const char *)q - buf.data()) + 1;
                p += (uint64_t)r;
            }
            return lo;
        }
        for (uint64_t e = hi; e > lo;) {                                   // backward from hi
            const uint64_t b = e - lo > buf.size() ? e - buf.size() : lo;
            const ssize_t r = pread(fd, buf.data(), e - b, (off_t)(offset + b));
            if (r != (ssize_t)(e - b)) break;
            const void *q = memrchr(buf.data(), '\n', (size_t)r);
            if (q) return b + (uint64_t)((const char *)q - buf.data()) + 1;
            e = b;
        }
        return lo;
    };
    const int rc = classify_pipelined(c, n, offset, want_hits, fetch, last_cut);
    close(fd);
    return rc;
}

extern "C" int svjg_get_stats(svjg_ctx *c, svjg_stats *out) {
    if (!c || !out) return SVJG_E_ARG;
    out->n_lines = c->hs().n_lines; out->n_deferred = c->total_deferred; out->n_hitrecs = c->hs().n_recs; out->non_ascii = c->hs().non_ascii;
    return 0;
}

extern "C" int svjg_get_defer_causes(svjg_ctx *c, uint64_t *out8) {
    if (!c || !out8) return SVJG_E_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = c->hs().cause[i];
    return 0;
}

extern "C" int svjg_input_error(svjg_ctx *c, int *cls, uint64_t *off) {
    if (!c || !cls || !off) return SVJG_E_ARG;
    if (c->hs().err == ~0ull) { *cls = SVJG_EXC_NONE; *off = 0; }
    else { *cls = (int)(c->hs().err & 7); *off = c->hs().err >> 3; }
    return 0;
}

extern "C" int svjg_get_counts(svjg_ctx *c, uint32_t *out, uint32_t n_slots) {
    if (!c || !c->have_counts || !out || n_slots != c->n_slots) return SVJG_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc0 = fetch_slot_counts(c); if (rc0) return rc0; }
    std::vector<unsigned long long> tmp(n_slots);
    if (n_slots) HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_counts, (uint64_t)n_slots * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n_slots; ++i) { out[2 * i] = (uint32_t)tmp[i]; out[2 * i + 1] = (uint32_t)(tmp[i] >> 32); }
    return 0;
}

extern "C" int svjg_set_counts(svjg_ctx *c, const uint32_t *in, uint32_t n_slots) {
    if (!c || !c->have_counts || !in || n_slots != c->n_slots) return SVJG_E_ARG;
    c->counts_in_slot = -1;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<unsigned long long> tmp(n_slots);
    for (uint32_t i = 0; i < n_slots; ++i) tmp[i] = (unsigned long long)in[2 * i] | ((unsigned long long)in[2 * i + 1] << 32);
    if (n_slots) HIPCHK(c, hipMemcpyAsync(c->d_counts, tmp.data(), (uint64_t)n_slots * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int svjg_get_host_lines(svjg_ctx *c, uint64_t *out, uint64_t cap, uint64_t *n) {
    if (!c || !n) return SVJG_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    *n = c->hs().n_host;
    const uint64_t have = c->hs().n_host < cap ? c->hs().n_host : cap;
    if (have && !out) return SVJG_E_ARG;
    if (have) HIPCHK(c, hipMemcpyAsync(out, c->d_host, have * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int svjg_get_hits(svjg_ctx *c, svjg_hitrec *out, uint64_t cap, uint64_t *n) {
    if (!c || !n) return SVJG_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t have = c->hs().n_recs < cap ? c->hs().n_recs : cap;
    if (have && !out) return SVJG_E_ARG;
    if (have) HIPCHK(c, hipMemcpyAsync(out, c->d_recs, have * sizeof(svjg_hitrec), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n = have;
    return 0;
}

// ---- multi-GPU ------------------------------------------------------------------------------------------

extern "C" int svjg_comm_unique_id(char *out128) {
    if (!out128) return SVJG_E_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return SVJG_E_RCCL;
    memcpy(out128, &id, 128);
    return 0;
}

extern "C" int svjg_comm_init(svjg_ctx *c, const char *id128, int n_ranks, int rank) {
    if (!c || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return SVJG_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, id, rank);
    if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); c->comm = nullptr; return SVJG_E_RCCL; }
    return 0;
}

extern "C" int svjg_comm_set_stream(svjg_ctx *c, int second_stream) {
    if (!c) return SVJG_E_ARG;
    if (c->run_inflight) { c->err = "svjg_comm_set_stream with a pass in flight"; return SVJG_E_ARG; }
    c->allreduce_second = second_stream != 0;
    return 0;
}

// the guard elements behind the count vector (svjg_pass.h) <- largest ref / alt field, and — st given: a fused pass — whether this
// rank's pass must be repeated (k_counts_guard)
static int launch_guard(svjg_ctx *c, unsigned long long *counts = nullptr, hipStream_t stream = nullptr, const DevStatus *st = nullptr) {
    if (!counts) counts = c->d_counts;
    if (!stream) stream = c->stream;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(counts + c->n_slots, 0, GUARD_WORDS * 8, stream));
    if (c->n_slots || st) {
        uint32_t grid = (c->n_slots + TPB - 1) / TPB;
        if (grid > 1024) grid = 1024;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(k_counts_guard, dim3(grid), dim3(TPB), 0, stream, counts, c->n_slots, st);
        HIPCHK(c, hipGetLastError());
    }
    return 0;
}
static int check_guard(svjg_ctx *c) {
    unsigned long long g[GUARD_WORDS] = {0, 0, 0};
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(g, c->d_counts + c->n_slots, sizeof g, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (pass_counts_overflowed(g[GUARD_MAX_REF], g[GUARD_MAX_ALT])) { c->err = "more than 2^32 informative alignments for one SV"; return SVJG_E_OVERFLOW; }
    return 0;
}

// the path's only collective: sum of the per-SV count vector (packed ref | alt << 32 as one u64 each, plus the two guard elements)
extern "C" int svjg_allreduce_counts(svjg_ctx *c) {
    if (!c || !c->have_counts) return SVJG_E_ARG;
    { const int rc0 = fetch_slot_counts(c); if (rc0) return rc0; }
    if (!c->comm) { c->err = "svjg_comm_init has not been called"; return SVJG_E_ARG; }
    int rc = launch_guard(c);
    if (rc) return rc;
    ncclResult_t r = ncclAllReduce(c->d_counts, c->d_counts, (size_t)c->n_slots + GUARD_WORDS, ncclUint64, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) { c->err = std::string("ncclAllReduce: ") + ncclGetErrorString(r); return SVJG_E_RCCL; }
    return check_guard(c);
}

// One process, several GPUs (the drop-in filter-alignments.py): one communicator per context from ncclCommInitAll, then the same
// all-reduce issued for every context between ncclGroupStart / ncclGroupEnd.  n == 1: only the overflow guard runs.
// svjg_comm_init_all may run in a thread of its own beside calls on the same contexts (svjg/filter.py: _CommInit), so what goes wrong in it
// is NOT written into a context's error string (another thread may be writing that): it has a message buffer of its own.
static std::mutex g_comm_err_mu;
static std::string g_comm_err;
static void comm_err_set(const std::string &m) { std::lock_guard<std::mutex> lk(g_comm_err_mu); g_comm_err = m; }
extern "C" int svjg_comm_error(char *out, uint64_t cap) {
    if (!out || !cap) return SVJG_E_ARG;
    std::lock_guard<std::mutex> lk(g_comm_err_mu);
    const size_t n = g_comm_err.size() < cap - 1 ? g_comm_err.size() : (size_t)cap - 1;
    memcpy(out, g_comm_err.data(), n); out[n] = 0;
    return 0;
}

extern "C" int svjg_comm_init_all(svjg_ctx *const *ctxs, int n) {
    if (!ctxs || n < 1 || n > 64) return SVJG_E_ARG;
    for (int i = 0; i < n; ++i) if (!ctxs[i]) return SVJG_E_ARG;
    if (n == 1) return 0;
    int devs[64];
    ncclComm_t comms[64];
    for (int i = 0; i < n; ++i) {
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j) if (devs[j] == devs[i]) { comm_err_set("svjg_comm_init_all: one context per device"); return SVJG_E_ARG; }
    }
    ncclResult_t r = ncclCommInitAll(comms, n, devs);
    if (r != ncclSuccess) { comm_err_set(std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); return SVJG_E_RCCL; }
    for (int i = 0; i < n; ++i) {
        if (ctxs[i]->comm) ncclCommDestroy(ctxs[i]->comm);
        ctxs[i]->comm = comms[i];
    }
    return 0;
}

extern "C" int svjg_allreduce_counts_all(svjg_ctx *const *ctxs, int n) {
    if (!ctxs || n < 1 || n > 64) return SVJG_E_ARG;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !ctxs[i]->have_counts || ctxs[i]->n_slots != ctxs[0]->n_slots) return SVJG_E_ARG;
        if (n > 1 && !ctxs[i]->comm) { ctxs[0]->err = "svjg_comm_init_all has not been called"; return SVJG_E_ARG; }
    }
    int rc;
    for (int i = 0; i < n; ++i) if ((rc = fetch_slot_counts(ctxs[i])) || (rc = launch_guard(ctxs[i]))) { if (i) ctxs[0]->err = ctxs[i]->err; return rc; }
    if (n > 1) {
        ncclResult_t r = ncclGroupStart();
        for (int i = 0; i < n && r == ncclSuccess; ++i) {
            svjg_ctx *c = ctxs[i];
            if (hipSetDevice(c->device) != hipSuccess) { ctxs[0]->err = "hipSetDevice"; ncclGroupEnd(); return SVJG_E_HIP; }
            r = ncclAllReduce(c->d_counts, c->d_counts, (size_t)c->n_slots + GUARD_WORDS, ncclUint64, ncclSum, c->comm, c->stream);
        }
        const ncclResult_t r2 = ncclGroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { ctxs[0]->err = std::string("ncclAllReduce (group): ") + ncclGetErrorString(r); return SVJG_E_RCCL; }
    }
    for (int i = 0; i < n; ++i) if ((rc = check_guard(ctxs[i]))) { if (i) ctxs[0]->err = ctxs[i]->err; return rc; }
    return 0;
}

// ---- genotypes -----------------------------------------------------------------------------------------

// table of log10(i!) in double-double for the binomial term, at least `upto` entries (kernels on the context's stream; no host wait)
static int build_logfact(svjg_ctx *c, uint32_t upto) {
    const uint32_t want = (upto + LF_BLOCK - 1) / LF_BLOCK * LF_BLOCK;
    hipFree(c->d_logfact); hipFree(c->d_bsum); c->d_logfact = nullptr; c->d_bsum = nullptr; c->logfact_n = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_logfact, (uint64_t)want * sizeof(dd)));
    HIPCHK(c, hipMalloc((void **)&c->d_bsum, (uint64_t)(want / LF_BLOCK) * sizeof(dd)));
    hipLaunchKernelGGL(k_logfact_local, dim3(want / LF_BLOCK), dim3(LF_BLOCK), 0, c->stream, c->d_logfact, c->d_bsum, want);
    hipLaunchKernelGGL(k_logfact_bsum, dim3(1), dim3(64), 0, c->stream, c->d_bsum, want / LF_BLOCK);
    hipLaunchKernelGGL(k_logfact_add, dim3(want / LF_BLOCK), dim3(LF_BLOCK), 0, c->stream, c->d_logfact, c->d_bsum, want);
    HIPCHK(c, hipGetLastError());
    c->logfact_n = want;
    return 0;
}

// results of all rows -> the pinned host block of the context: [ pl 24 | raw 8 | gt 1 | done 1 | boundary 1 ] x n_rows
static int genotype_rows(svjg_ctx *c, const uint8_t *sv_type, const uint32_t *slot, const uint8_t *ok, uint64_t n_rows,
                         uint32_t min_support, double err) {
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc0 = fetch_slot_counts(c); if (rc0) return rc0; }
    // one device block and its pinned host twin: [ pl 24 | raw 8 | gt 1 | done 1 | boundary 1 ] n rows of output, max_n, then
    // [ slot 4 | type 1 | ok 1 ] n rows of input -> ONE copy in and ONE copy out per call whatever the number of arrays
    const uint64_t out_bytes = n_rows * 35, maxn_off = (out_bytes + 7) & ~7ull, in_off = maxn_off + 8, in_bytes = n_rows * 6;
    const uint64_t total = in_off + in_bytes + 64;
    int rc = ensure(c, &c->d_rows, &c->rows_cap, total, 1, false);
    if (rc) return rc;
    if (total > c->h_rows_cap) {
        if (c->h_rows) hipHostFree(c->h_rows);
        c->h_rows = nullptr; c->h_rows_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->h_rows, total, hipHostMallocDefault));
        c->h_rows_cap = total;
    }
    uint8_t *base = (uint8_t *)c->d_rows, *hb = (uint8_t *)c->h_rows;
    int64_t *d_pl = (int64_t *)base;
    uint32_t *d_raw = (uint32_t *)(base + n_rows * 24);
    uint8_t *d_gt = base + n_rows * 32, *d_done = base + n_rows * 33;
    unsigned int *d_maxn = (unsigned int *)(base + maxn_off);
    uint32_t *d_slot = (uint32_t *)(base + in_off);
    uint8_t *d_type = base + in_off + n_rows * 4, *d_ok = base + in_off + n_rows * 5;
    memcpy(hb + in_off, slot, n_rows * 4); memcpy(hb + in_off + n_rows * 4, sv_type, n_rows); memcpy(hb + in_off + n_rows * 5, ok, n_rows);
    HIPCHK(c, hipMemcpyAsync(base + in_off, hb + in_off, in_bytes, hipMemcpyHostToDevice, c->stream));
    GenoArgs a{};
    a.counts = c->d_counts; a.sv_type = d_type; a.slot = d_slot; a.ok = d_ok; a.n_rows = n_rows; a.min_support = min_support;
    a.l_ok = log10(1.0 - err); a.l_err = log10(err); a.l_half = log10(1.0 / 2.0);     // host libm, as CPython's math.log10
    a.gt = d_gt; a.pl = d_pl; a.raw = d_raw; a.genotyped = d_done; a.boundary = base + n_rows * 34; a.max_n = d_maxn; a.n_slots = c->n_slots;
    c->geno_rows = n_rows;
    const uint32_t grid = (uint32_t)((n_rows + TPB - 1) / TPB);
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    // One pass with the log10(i!) table at hand; the kernel reports the largest n = ref + alt it met beyond the table, and
    // only then (first call, or a deeper sample than ever before) the table is rebuilt and the pass repeated.
    uint32_t grow_to = 65536;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (c->logfact_n == 0 && (rc = build_logfact(c, grow_to))) return rc;
        a.logfact = c->d_logfact; a.logfact_n = c->logfact_n;
        HIPCHK(c, hipMemsetAsync(d_maxn, 0, 8, c->stream));
        hipLaunchKernelGGL(k_genotype, dim3(grid), dim3(TPB), 0, c->stream, a);
        HIPCHK(c, hipGetLastError());
        if (attempt == 0) HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
        HIPCHK(c, hipMemcpyAsync(hb, base, maxn_off + 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (*(const unsigned int *)(hb + maxn_off + 4)) { c->err = "slot out of range"; return SVJG_E_ARG; }   // (checked by the kernel, row by row)
        const unsigned int max_n = *(const unsigned int *)(hb + maxn_off);
        if (max_n == 0) break;                               // every row found its binomial term
        if (attempt == 1) { c->err = "log10(i!) table could not be sized"; return SVJG_E_HIP; }
        grow_to = max_n + 1 + 1024;
        c->logfact_n = 0;                                    // rebuild, sized by max_n
    }
    HIPCHK(c, hipEventElapsedTime(&c->ms_geno, c->ev[4], c->ev[5]));
    return 0;
}

extern "C" int svjg_genotype(svjg_ctx *c, const uint8_t *sv_type, const uint32_t *slot, const uint8_t *ok, uint64_t n_rows,
                             uint32_t min_support, double err, uint8_t *gt, int64_t *pl, uint32_t *raw, uint8_t *genotyped) {
    if (!c || !c->have_counts) return SVJG_E_ARG;
    if (n_rows == 0) return 0;
    if (!sv_type || !slot || !ok || !gt || !pl || !raw || !genotyped) return SVJG_E_ARG;
    const int rc = genotype_rows(c, sv_type, slot, ok, n_rows, min_support, err);
    if (rc) return rc;
    const uint8_t *hb = (const uint8_t *)c->h_rows;
    memcpy(pl, hb, n_rows * 24); memcpy(raw, hb + n_rows * 24, n_rows * 8);
    memcpy(gt, hb + n_rows * 32, n_rows); memcpy(genotyped, hb + n_rows * 33, n_rows);
    return 0;
}

// The same, results left where the device wrote them: the four pointers look into the context's pinned host block and stay
// valid until the next svjg_genotype / svjg_genotype_view / svjg_destroy on this context.
extern "C" int svjg_genotype_view(svjg_ctx *c, const uint8_t *sv_type, const uint32_t *slot, const uint8_t *ok, uint64_t n_rows,
                                  uint32_t min_support, double err, const uint8_t **gt, const int64_t **pl, const uint32_t **raw,
                                  const uint8_t **genotyped) {
    if (!c || !c->have_counts || !gt || !pl || !raw || !genotyped) return SVJG_E_ARG;
    *gt = nullptr; *pl = nullptr; *raw = nullptr; *genotyped = nullptr;
    if (n_rows == 0) return 0;
    if (!sv_type || !slot || !ok) return SVJG_E_ARG;
    const int rc = genotype_rows(c, sv_type, slot, ok, n_rows, min_support, err);
    if (rc) return rc;
    const uint8_t *hb = (const uint8_t *)c->h_rows;
    *pl = (const int64_t *)hb; *raw = (const uint32_t *)(hb + n_rows * 24);
    *gt = hb + n_rows * 32; *genotyped = hb + n_rows * 33;
    return 0;
}

// ---- the whole pass in one call, one host wait; two passes in flight ------------------------------------------------------
// svjg_set_rows leaves the VCF rows' three input arrays on the device.  svjg_run_begin enqueues a pass — back to back on the
// context's stream, no host round trip in between: zero counts, classify the resident text (the exact-path kernel takes the
// number of deferred lines from the device), the count all-reduce when the context has a communicator, genotype every row — and,
// on a second stream behind an event, the copy of the results (PLs as 32-bit integers) and of the pass's status block into pinned
// host memory.  svjg_run_end waits for the OLDEST pass in flight, checks what the host used to decide between the kernels
// (a list that overflowed, more deferred lines than one wave per line is good for, a log10(i!) table too short: the pass is then
// repeated the slow way) and hands out its results.  Up to two passes may be in flight: the results of pass k travel over PCIe
// while pass k + 1 computes.  svjg_run_resident = begin + end.
// host block (pinned, mapped into the device: the genotype kernel writes its results straight into it — they cross PCIe as they
// are produced, no copy kernel competes with the next pass —): pl32, raw, gt, flags, boundary, then the tail; device block: the
// tail (max_n, the pass's status block, the guard words: written by atomics, copied to the host block's tail in one small copy), pl64
struct RunLayout { uint64_t pl32, raw, gt, flags, boundary, h_tail, out_bytes;  uint64_t maxn, status, guard, tail_bytes, pl64, total; };
static RunLayout run_layout(uint64_t n) {
    RunLayout L; uint64_t o = 0;
    L.pl32 = o; o += n * 12; L.raw = o; o += n * 8; L.gt = o; o += n; L.flags = o; o += n; L.boundary = o; o += n; o = (o + 63) & ~63ull; L.h_tail = o;
    uint64_t d = 0;
    L.maxn = d; d += 8; L.status = d; d += (sizeof(DevStatus) + 7) & ~7ull; L.guard = d; d += GUARD_WORDS * 8; L.tail_bytes = d;
    L.out_bytes = L.h_tail + L.tail_bytes;
    d = (d + 63) & ~63ull; L.pl64 = d; d += n * 24; L.total = d + 64;
    return L;
}

extern "C" int svjg_set_rows(svjg_ctx *c, const uint8_t *sv_type, const uint32_t *slot, const uint8_t *ok, uint64_t n_rows) {
    if (!c || (n_rows && (!sv_type || !slot || !ok))) return SVJG_E_ARG;
    if (c->run_inflight) { c->err = "svjg_set_rows with a pass in flight"; return SVJG_E_ARG; }
    if (!c->have_counts) { c->err = "svjg_set_rows needs the count vector (svjg_load_graph / svjg_alloc_counts first)"; return SVJG_E_ARG; }
    { const int rc0 = fetch_slot_counts(c); if (rc0) return rc0; }
    HIPCHK(c, hipSetDevice(c->device));
    const RunLayout L = run_layout(n_rows);
    int rc;
    if ((rc = ensure(c, &c->d_run_in, &c->d_run_in_cap, n_rows * 6 + 64, 1, false))) return rc;
    for (int k = 0; k < 2; ++k) {
        svjg_ctx::RunSlot &r = c->run[k];
        if ((rc = ensure(c, &r.d, &r.d_cap, L.total, 1, false))) return rc;
        if (L.out_bytes > r.h_cap) {
            if (r.h) hipHostFree(r.h);
            r.h = nullptr; r.h_cap = 0;
            HIPCHK(c, hipHostMalloc(&r.h, L.out_bytes, hipHostMallocMapped));
            r.h_cap = L.out_bytes;
            HIPCHK(c, hipHostGetDevicePointer(&r.h_dev, r.h, 0));
        }
        if ((rc = ensure(c, (void **)&r.counts, &r.counts_cap, (uint64_t)c->n_slots + GUARD_WORDS, sizeof(unsigned long long), false))) return rc;
        for (auto &e : r.ev) if (!e) HIPCHK(c, hipEventCreate(&e));
        if (!r.computed) HIPCHK(c, hipEventCreate(&r.computed));          // (with a time stamp: it also ends the exact-path kernels' interval, below)
        if (!r.copied) HIPCHK(c, hipEventCreateWithFlags(&r.copied, hipEventDisableTiming));
    }
    if (!c->copy_stream) {                                    // (the lowest priority there is: what runs on it must not take issue slots from the classify kernel)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        HIPCHK(c, hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, least));
    }
    uint8_t *in = (uint8_t *)c->d_run_in;
    if (n_rows) {
        HIPCHK(c, hipMemcpyAsync(in, slot, n_rows * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(in + n_rows * 4, sv_type, n_rows, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(in + n_rows * 5, ok, n_rows, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));               // (the caller's arrays may go away)
    c->run_rows = n_rows; c->have_rows = true;
    return 0;
}

static GenoArgs run_geno_args(svjg_ctx *c, const svjg_ctx::RunSlot &r, const RunLayout &L, uint32_t min_support, double err, const unsigned long long *counts) {
    uint8_t *base = (uint8_t *)r.d, *hostb = (uint8_t *)r.h_dev;
    const uint8_t *in = (const uint8_t *)c->d_run_in;
    GenoArgs a{};
    a.counts = counts; a.slot = (const uint32_t *)in; a.sv_type = in + c->run_rows * 4; a.ok = in + c->run_rows * 5; a.n_rows = c->run_rows;
    a.min_support = min_support;
    a.l_ok = log10(1.0 - err); a.l_err = log10(err); a.l_half = log10(1.0 / 2.0);     // host libm, as CPython's math.log10
    a.gt = hostb + L.gt; a.pl = (int64_t *)(base + L.pl64); a.raw = (uint32_t *)(hostb + L.raw); a.genotyped = hostb + L.flags;
    a.pl32 = (int32_t *)(hostb + L.pl32); a.boundary = hostb + L.boundary; a.max_n = (unsigned int *)(base + L.maxn); a.n_slots = c->n_slots;
    a.logfact = c->d_logfact; a.logfact_n = c->logfact_n;
    return a;
}

extern "C" int svjg_run_begin(svjg_ctx *c, uint64_t base_offset, uint32_t min_support, double err) {
    if (!c) return SVJG_E_ARG;
    if (!c->have_graph || !c->have_gaf) { c->err = "svjg_run_begin needs a graph and an uploaded GAF"; return SVJG_E_ARG; }
    if (!c->have_rows) { c->err = "svjg_set_rows has not been called"; return SVJG_E_ARG; }
    if (c->run_inflight >= 2) { c->err = "two passes are in flight: svjg_run_end first"; return SVJG_E_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t n = c->gaf_bytes, n_rows = c->run_rows;
    const RunLayout L = run_layout(n_rows);
    svjg_ctx::RunSlot &r = c->run[c->run_head];
    int rc;
    if ((rc = ensure(c, (void **)&c->d_deferred, &c->deferred_cap, deferred_want(c, n), sizeof(uint64_t), false))) return rc;
    if ((rc = ensure(c, (void **)&c->d_host, &c->host_cap, 4096, sizeof(uint64_t), false))) return rc;
    if (c->logfact_n == 0 && (rc = build_logfact(c, 65536))) return rc;
    r.base_offset = base_offset; r.min_support = min_support; r.err = err;
    uint8_t *base = (uint8_t *)r.d;
    DevStatus *d_st = (DevStatus *)(base + L.status);             // (the pass's own status block: its tail goes to the host in one small copy)
    if (r.counts_cap < (uint64_t)c->n_slots + GUARD_WORDS) { c->err = "svjg_set_rows must follow svjg_load_graph"; return SVJG_E_ARG; }
    GenoArgs ga = run_geno_args(c, r, L, min_support, err, r.counts);
    {
        const uint64_t words = (uint64_t)c->n_slots + GUARD_WORDS;
        uint32_t rg = (uint32_t)((words + TPB - 1) / TPB);
        if (rg > 1024) rg = 1024;
        hipLaunchKernelGGL(k_step_reset, dim3(rg), dim3(TPB), 0, c->stream, r.counts, words, d_st, ga.max_n);
    }
    const uint64_t max_blocks = (uint64_t)c->n_cu * 4, wave_limit = 16 * max_blocks;
    if (n) {
        ClassifyArgs a{};
        uint32_t grid = 0;
        size_t lds = 0;
        main_launch_setup(c, 0, n, base_offset, 0, a, grid, lds);
        if (!a.long_pre) { c->err = "no memory for the workers' scratch words"; return SVJG_E_NOMEM; }
        a.st = d_st; a.counts = r.counts;
        HIPCHK(c, hipEventRecord(r.ev[0], c->stream));
        hipLaunchKernelGGL(k_classify_main, dim3(grid), dim3(WG), lds, c->stream, a);
        HIPCHK(c, hipEventRecord(r.ev[1], c->stream));
        // the exact path, the number of deferred lines read on the device: one wave per line for up to wave_limit lines, one lane per
        // line beyond it (each kernel works only when the number lies in its range: a few microseconds otherwise).  So whatever a shard
        // defers is counted before the pass's all-reduce; only a LIST that overflowed makes the pass repeat (svjg_pass.h).  (One block per CU
        // for the wave kernel: three would triple its rate and cost every pass that defers nothing 4 us; the lane kernel's 2 304 idle blocks
        // cost such a pass 10 us, 0.85 % of the headline step — the price of never repeating a pass for deferred lines: gpurun j32 / j33.)
        hipLaunchKernelGGL(k_classify_slow_wave, dim3((uint32_t)c->n_cu), dim3(SLOW_TPB), 0, c->stream, a, SLOW_ASK_DEVICE, 0ull, wave_limit);
        const uint64_t lane_blocks = (uint64_t)c->n_cu * lane_blocks_per_cu(c);   // (here as many as the CUs hold at a time: idle blocks cost every pass)
        hipLaunchKernelGGL(k_classify_slow, dim3((uint32_t)lane_blocks), dim3(SLOW_TPB), 0, c->stream, a, SLOW_ASK_DEVICE, wave_limit, ~0ull - 1);
        HIPCHK(c, hipGetLastError());
        // (no event of their own behind the two: `computed`, a few lines down, ends their interval — an event record is a barrier packet,
        //  6 us of every pass: profiles/r05/experiments/pass_overhead.txt)
    }
    // The count all-reduce of this pass — every rank issues its collectives in the same order, one per pass — runs on the compute
    // stream, between this pass's kernels and the next pass's: the classify kernel fills every CU (fourteen workers take 126 of a
    // CU's 128 LDS granules and all the registers of two SIMDs), so a collective's workgroups find no room beside it and one queued
    // on the second stream would only start when the NEXT pass's classify kernel drains — and hold back this pass's results, which
    // the host waits for before it may enqueue the pass after.  (SVJG_ALLREDUCE_STREAM=second puts it there all the same: for a
    // measurement on a multi-GPU box.)  Then, behind an event, on the second (low-priority) stream, the genotypes: one wave per CU
    // fits beside the classify workers; the kernel writes its results straight into the pinned host block — they cross PCIe as they
    // are produced —, which takes ~45 us for 100 k rows.  Meanwhile the compute stream already runs the next pass (that one zeroes
    // and fills ITS count vector).
    static const bool env_second = [] { const char *e = getenv("SVJG_ALLREDUCE_STREAM"); return e && !strcmp(e, "second"); }();
    const bool allreduce_second = env_second || c->allreduce_second;
    auto reduce_on = [&](hipStream_t st) -> int {
        int rc2 = launch_guard(c, r.counts, st, d_st);                  // (guard word 2: "this rank's lists overflowed")
        if (rc2) return rc2;
        ncclResult_t nr = ncclAllReduce(r.counts, r.counts, (size_t)c->n_slots + GUARD_WORDS, ncclUint64, ncclSum, c->comm, st);
        if (nr != ncclSuccess) { c->err = std::string("ncclAllReduce: ") + ncclGetErrorString(nr); return SVJG_E_RCCL; }
        HIPCHK(c, hipMemcpyAsync(base + L.guard, r.counts + c->n_slots, GUARD_WORDS * 8, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    r.slow_end_own = false;
    if (c->comm && !allreduce_second) {
        // (the exact-path kernels' interval must not hold the collective: under a communicator it gets an end event of its own — 6 us of a
        //  multi-GPU pass, none of a one-GPU pass, whose interval `computed` ends)
        if (r.had_text) { HIPCHK(c, hipEventRecord(r.ev[2], c->stream)); r.slow_end_own = true; }
        if ((rc = reduce_on(c->stream))) return rc;
    }
    HIPCHK(c, hipEventRecord(r.computed, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, r.computed, 0));
    if (c->comm && allreduce_second && (rc = reduce_on(c->copy_stream))) return rc;
    if (n_rows) {
        HIPCHK(c, hipEventRecord(r.ev[4], c->copy_stream));
        // one wave per CU (see k_genotype); SVJG_GENO_GRID / SVJG_GENO_BLOCK: measurement only
        uint32_t gb = 64, gg = (uint32_t)c->n_cu;
        { const char *e = getenv("SVJG_GENO_BLOCK"); if (e && atoi(e) > 0) gb = (uint32_t)atoi(e); }
        { const char *e = getenv("SVJG_GENO_GRID"); if (e && atoi(e) > 0) gg = (uint32_t)atoi(e); }
        if (gb > TPB) gb = TPB;
        if ((uint64_t)gg * gb > n_rows + gb) gg = (uint32_t)((n_rows + gb - 1) / gb);
        hipLaunchKernelGGL(k_genotype, dim3(gg), dim3(gb), 0, c->copy_stream, ga);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(r.ev[5], c->copy_stream));
    }
    // the pass's tail: max_n, status, guard words
    HIPCHK(c, hipMemcpyAsync((uint8_t *)r.h + L.h_tail, r.d, L.tail_bytes, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipEventRecord(r.copied, c->copy_stream));
    c->counts_in_slot = c->run_head;
    r.had_text = n != 0;
    c->run_head ^= 1; ++c->run_inflight;
    return 0;
}

extern "C" int svjg_run_end(svjg_ctx *c, const uint8_t **gt, const int32_t **pl, const uint32_t **raw, const uint8_t **flags, const uint8_t **boundary) {
    if (!c || !gt || !pl || !raw || !flags || !boundary) return SVJG_E_ARG;
    *gt = nullptr; *pl = nullptr; *raw = nullptr; *flags = nullptr; *boundary = nullptr;
    if (!c->run_inflight) { c->err = "no pass in flight"; return SVJG_E_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    svjg_ctx::RunSlot &r = c->run[c->run_tail];
    c->run_tail ^= 1; --c->run_inflight;
    const uint64_t n = c->gaf_bytes, n_rows = c->run_rows;
    const RunLayout L = run_layout(n_rows);
    uint8_t *hb = (uint8_t *)r.h;
    int rc;
    HIPCHK(c, hipEventSynchronize(r.copied));                              // the pass's one host wait
    const uint8_t *tail = hb + L.h_tail;
    c->hs() = *(const DevStatus *)(tail + L.status);
    // ---- what the host would have decided in between ----
    c->ms_slow = 0;
    if (r.had_text) {
        HIPCHK(c, hipEventElapsedTime(&c->ms_main, r.ev[0], r.ev[1]));
        if (c->hs().n_deferred) HIPCHK(c, hipEventElapsedTime(&c->ms_slow, r.ev[1], r.slow_end_own ? r.ev[2] : r.computed));   // (the exact-path kernels alone: never the all-reduce)
    }
    if (n_rows) HIPCHK(c, hipEventElapsedTime(&c->ms_geno, r.ev[4], r.ev[5]));
    bool again = false;
    const unsigned long long *redo_counts = r.counts;             // (a repeated genotype pass reads the slot's vector, or d_counts after the step-by-step fallback)
    const unsigned long long *gd = (const unsigned long long *)(tail + L.guard);   // (meaningful under a communicator: the sums over the ranks)
    if (pass_repeats(c->comm != nullptr, c->hs().overflow, gd[GUARD_REPEAT])) {
        // a list of this rank — or, under a communicator, of ANY rank (svjg_pass.h: the ranks decide together, so all of them issue the
        // one more all-reduce below) — was too short: the pass again, step by step (classify_range sizes the lists, picks the exact-path
        // kernel and retries), behind whatever is enqueued already.  The all-reduce is issued even when this rank's text turns out to be
        // malformed: its peers are waiting in theirs.
        if ((rc = svjg_reset_counts(c))) return rc;
        rc = classify_range(c, 0, n, r.base_offset, 0);
        if (rc && rc != SVJG_E_INPUT) return rc;
        if (c->comm) { const std::string keep = c->err; const int rc2 = svjg_allreduce_counts(c); if (rc2 && !rc) return rc2; if (rc) c->err = keep; }
        if (rc) return rc;
        again = true; redo_counts = c->d_counts;
    } else {
        c->total_deferred = c->hs().n_deferred;
        if (c->hs().err != ~0ull) return SVJG_E_INPUT;
        if (c->comm && pass_counts_overflowed(gd[GUARD_MAX_REF], gd[GUARD_MAX_ALT])) { c->err = "more than 2^32 informative alignments for one SV"; return SVJG_E_OVERFLOW; }
    }
    // (either way: lines the kernels set aside for the host are neither counted nor fatal — the caller has to know)
    if (c->hs().n_host) { c->err = "the text holds lines only the host can decide (non-ASCII digits in a decimal column: svjg_get_host_lines); classify it with svjg_classify"; return SVJG_E_ARG; }
    if (n_rows) {
        for (int attempt = 0;; ++attempt) {
            if (!again) {
                if (*(const unsigned int *)(tail + L.maxn + 4)) { c->err = "slot out of range"; return SVJG_E_ARG; }
                const unsigned int max_n = *(const unsigned int *)(tail + L.maxn);
                if (max_n == 0) break;                           // every row found its binomial term
                if (attempt == 2) { c->err = "log10(i!) table could not be sized"; return SVJG_E_HIP; }
                // (a pass already enqueued behind this one still uses the old table: it must drain before the table is replaced)
                HIPCHK(c, hipStreamSynchronize(c->stream));
                HIPCHK(c, hipStreamSynchronize(c->copy_stream));
                if ((rc = build_logfact(c, max_n + 1 + 1024))) return rc;
            }
            again = false;
            GenoArgs ga = run_geno_args(c, r, L, r.min_support, r.err, redo_counts);
            HIPCHK(c, hipMemsetAsync(ga.max_n, 0, 8, c->stream));
            hipLaunchKernelGGL(k_genotype, dim3((uint32_t)((n_rows + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, ga);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(hb + L.h_tail + L.maxn, (uint8_t *)r.d + L.maxn, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    *pl = (const int32_t *)(hb + L.pl32); *raw = (const uint32_t *)(hb + L.raw); *gt = hb + L.gt; *flags = hb + L.flags; *boundary = hb + L.boundary;
    return 0;
}

extern "C" int svjg_run_resident(svjg_ctx *c, uint64_t base_offset, uint32_t min_support, double err,
                                 const uint8_t **gt, const int32_t **pl, const uint32_t **raw, const uint8_t **flags, const uint8_t **boundary) {
    if (!c || !gt || !pl || !raw || !flags || !boundary) return SVJG_E_ARG;
    if (c->run_inflight) { c->err = "svjg_run_resident with a pass in flight (svjg_run_end first)"; return SVJG_E_ARG; }
    const int rc = svjg_run_begin(c, base_offset, min_support, err);
    if (rc) return rc;
    return svjg_run_end(c, gt, pl, raw, flags, boundary);
}

// which rows of the last svjg_genotype / svjg_genotype_view call lie so close to a PL's integer boundary that the caller should
// recompute them with the reference's own arithmetic (predict-genotype.py:313: log10 of an exact big integer)
extern "C" int svjg_genotype_boundary(svjg_ctx *c, uint8_t *out, uint64_t n_rows) {
    if (!c || (n_rows && !out) || n_rows != c->geno_rows || !c->h_rows) return SVJG_E_ARG;
    memcpy(out, (const uint8_t *)c->h_rows + n_rows * 34, n_rows);
    return 0;
}

extern "C" int svjg_last_kernel_ms(svjg_ctx *c, float *m, float *s, float *g) {
    if (!c) return SVJG_E_ARG;
    if (m) *m = c->ms_main;
    if (s) *s = c->ms_slow;
    if (g) *g = c->ms_geno;
    return 0;
}

extern "C" int svjg_copy_rate(svjg_ctx *c, uint64_t n_bytes, double *copy_gb_per_s, double *read_gb_per_s) {
    if (!c || n_bytes < (1u << 20)) return SVJG_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    n_bytes &= ~4095ull;
    uint4 *src = nullptr, *dst = nullptr;
    if (hipMalloc((void **)&src, n_bytes) != hipSuccess || hipMalloc((void **)&dst, n_bytes) != hipSuccess) { hipFree(src); c->err = "svjg_copy_rate: no memory for the two buffers"; return SVJG_E_NOMEM; }
    struct Free { uint4 *a, *b; ~Free() { hipFree(a); hipFree(b); } } fr{src, dst};
    HIPCHK(c, hipMemsetAsync(src, 0x5A, n_bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(dst, 0, n_bytes, c->stream));
    const uint32_t grid = (uint32_t)c->n_cu * 16;
    for (int what = 0; what < 2; ++what) {
        double *out = what ? read_gb_per_s : copy_gb_per_s;
        if (!out) continue;
        float best = 0;
        for (int i = 0; i < 4; ++i) {                          // (the first run warms up)
            HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
            if (what) hipLaunchKernelGGL(k_read16, dim3(grid), dim3(TPB), 0, c->stream, dst, src, n_bytes / 16);
            else hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(TPB), 0, c->stream, dst, src, n_bytes / 16);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
            if (i && (best == 0 || ms < best)) best = ms;
        }
        *out = best > 0 ? (what ? 1.0 : 2.0) * (double)n_bytes / (best * 1e-3) / 1e9 : 0.0;
    }
    return 0;
}

extern "C" int svjg_sync(svjg_ctx *c) {
    if (!c) return SVJG_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
