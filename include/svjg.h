/* svjg.h — C ABI of libsvjg_hip.so: the MI355X (gfx950) implementation of SVJedi-graph's
 * alignment-classification + genotype-likelihood hot path.
 *
 * The reference (SandraLouise/SVJedi-graph) has no FFI for this path: its boundary is two Python
 * scripts run as shell commands (svjedi-graph.py:114-115, :124-125).  The drop-in scripts in
 * svjedi-graph_amd/ keep that command-line / file contract and call the entry points below through
 * ctypes (svjedi-graph_amd/svjg/capi.py).  Each entry point names the reference code it replaces.
 *
 * Conventions: every function returns 0 on success and a negative SVJG_E_* code on failure (message via
 * svjg_last_error).  The caller owns every host buffer it passes in or out; the library owns all device
 * memory.  Calls on one context must be serialised by the caller.  All structs are fixed-width
 * little-endian PODs.  No exceptions or aborts cross the boundary.  There is no CPU fallback: without a
 * GPU svjg_init fails with SVJG_E_NO_DEVICE.
 */
#ifndef SVJG_H
#define SVJG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVJG_ABI_VERSION 1

/* error codes (negative returns) */
#define SVJG_E_NO_DEVICE   (-1)   /* no usable HIP device */
#define SVJG_E_HIP         (-2)   /* a HIP runtime call failed */
#define SVJG_E_ARG         (-3)   /* bad argument / call order */
#define SVJG_E_NOMEM       (-4)
#define SVJG_E_RCCL        (-5)
#define SVJG_E_IO          (-6)   /* the GAF file could not be opened / read (svjg_gaf_upload_file) */
/* The input made the reference raise (it would exit 1, svjedi-graph.py:117-118).  svjg_input_error()
 * tells which Python exception and at which byte offset of the GAF. */
#define SVJG_E_INPUT       (-10)
#define SVJG_E_OVERFLOW    (-12)  /* 2^32 or more informative alignments for one SV (the count fields are 32 bits wide) */

/* exception class the reference would have died with (svjg_input_error) */
#define SVJG_EXC_NONE            0
#define SVJG_EXC_VALUE_ERROR     1   /* int()/float() of a column, too few columns (filter-alignments.py:185-194) */
#define SVJG_EXC_INDEX_ERROR     2   /* empty path column, unoriented multi-node path, node name without '-' */
#define SVJG_EXC_KEY_ERROR       3   /* alt node missing from the GFA (filter-alignments.py:346) */
#define SVJG_EXC_ZERO_DIVISION   4   /* Alen == 0 without an id:f: tag (filter-alignments.py:196) */
/* Not an exception: int() / float() / str.rstrip() of the reference also take non-ASCII digits and blanks (Unicode categories Nd,
 * Zs, ...), which the kernels do not read.  A line whose decimal column or id:f: value fails the ASCII rules AND holds a byte
 * >= 0x80 is neither counted nor an error: its offset goes to a list (svjg_get_host_lines) and the host decides with Python's
 * own int() / float() (svjedi-graph_amd/svjg/filter.py: resolve_host_lines), resubmitting an ASCII spelling of the line if the
 * reference accepts it. */
#define SVJG_EXC_ASK_HOST        5
/* -O / --dover given (SVJG_GRAPH_DOVER_LIST): argparse hands the reference a LIST, and the first comparison with it — the left
 * overlap of the first link that has a candidate SV, after the node lengths of that sum were looked up — dies with
 * TypeError (filter-alignments.py:52-57, :88, :153 -> :269). */
#define SVJG_EXC_TYPE_ERROR      6

typedef struct svjg_ctx svjg_ctx;

/* ---- graph tables (built on the host by svjedi-graph_amd/svjg/graph.py) ---------------------------
 * They replace d_link_sv (filter-alignments.py:95-98), alt_node_len (:103-113) and the node-name
 * arithmetic of get_node_start/end/len (:328-349).
 *
 * Nodes are sorted by `key` = chrom_idx << 48 | pos << 16 | kind << 15 | cnt
 *   reference node "chrom:start-end": pos = start, kind = 0, cnt = 0, aux = end
 *   alt node       "chrom:pos.cnt"  : pos = pos,   kind = 1, cnt = cnt, aux = sequence length
 *                                     (SVJG_LEN_UNKNOWN if the GFA has no S-line for it)
 * row   = first entry of this node's links in the edge array (rows are contiguous, node n's links are
 *         [row(n), row(n+1)); the node array carries one sentinel element at the end)
 * flags = SVJG_NODE_HAZARD if the node's name is a proper substring of another node name of the graph
 *         (the strand quirk of filter-alignments.py:206 can then depend on the other nodes of the path)
 */
#define SVJG_LEN_UNKNOWN 0xFFFFFFFFu
#define SVJG_NODE_HAZARD 1u

typedef struct {
    uint64_t key;
    uint32_t aux;
    uint32_t row;      /* bit 31 = SVJG_NODE_HAZARD, bits 0..30 = row */
} svjg_node;

/* One directed link query (left node = the row it sits in).  The host stores, for every key K of
 * *_svs_edges.json and its reversed form R (filter-alignments.py:221-225), the concatenation
 * d[K] ++ d[R] under K and d[R] ++ d[K] under R, so that ONE probe per path step returns exactly what
 * the reference collects with its two dictionary probes (:141-153), multiplicities included.
 *   meta bit 0 = strand of the left node  (1 = '-'), bit 1 = strand of the right node, bits 2.. = n_hits
 *   n_hits <= 2: h0, h1 are the hits;  n_hits > 2: h0 = first index into the hit array
 *   a hit is slot << 1 | allele
 */
typedef struct {
    uint32_t right;
    uint32_t meta;
    uint32_t h0;
    uint32_t h1;
} svjg_edge;

/* chromosome dictionary: names concatenated in `chrom_names`, chrom i = [chrom_off[i], chrom_off[i+1]);
 * chrom_node_lo[i] .. chrom_node_lo[i+1] is its node range. */
typedef struct {
    const svjg_node *nodes;        uint64_t n_nodes;       /* without the sentinel; nodes[n_nodes] is the sentinel */
    const svjg_edge *edges;        uint64_t n_edges;
    const uint32_t  *hits;         uint64_t n_hits;        /* overflow hit lists */
    const char      *chrom_names;  const uint32_t *chrom_off;  const uint32_t *chrom_node_lo;  uint32_t n_chrom;
    uint32_t         n_slots;      /* number of distinct sv_id keys = length of the count vector */
    uint32_t         d_over;       /* minimum breakpoint overlap, 100 (filter-alignments.py:56,88) */
    uint32_t         flags;        /* SVJG_GRAPH_* */
} svjg_graph;

#define SVJG_GRAPH_ALL_SLOW 1u     /* route every alignment through the exact string path (debug / odd names) */
/* The reference was given -O: d_over is a list there, and the first link with a candidate SV whose left-overlap sum can be formed
 * raises TypeError (SVJG_EXC_TYPE_ERROR).  Every line takes the exact path (the flag implies SVJG_GRAPH_ALL_SLOW); lines in front
 * of that link are classified — and may raise — as usual, and a file without such a link is written as the reference writes it. */
#define SVJG_GRAPH_DOVER_LIST 16u

/* One informative (alignment, SV) pair: what filter-alignments.py:163-166 appends, run-length encoded —
 * n_ref / n_alt = how many times the line is appended to the ref / alt list of that SV. */
typedef struct {
    uint64_t line_start;           /* byte offset of the line in the GAF given to svjg_classify */
    uint32_t slot;
    uint16_t n_ref;
    uint16_t n_alt;
} svjg_hitrec;

typedef struct {
    uint64_t n_lines;              /* GAF lines seen */
    uint64_t n_deferred;           /* lines that took the exact string path */
    uint64_t n_hitrecs;
    uint64_t non_ascii;            /* 1 if any byte >= 0x80 was seen (host must validate UTF-8 like the reference's text-mode read) */
} svjg_stats;

/* ---- lifecycle ---------------------------------------------------------------------------------- */
int  svjg_abi_version(void);
int  svjg_device_count(void);
int  svjg_init(int device, svjg_ctx **out);
void svjg_destroy(svjg_ctx *ctx);
const char *svjg_last_error(const svjg_ctx *ctx);          /* ctx may be NULL: error of the last failed svjg_init */

/* ---- graph (filter-alignments.py:95-113) ---------------------------------------------------------- */
int svjg_load_graph(svjg_ctx *ctx, const svjg_graph *g);
/* The lookup tables the kernels use are derived from `g` on the host (seconds for a graph of 500 k SVs) and kept for the next
 * svjg_load_graph of the same graph in this process (one context per GPU: they are built once); this drops the kept copy. */
void svjg_release_host_tables(void);

/* ---- alignments (filter-alignments.py:123-166) ----------------------------------------------------
 * svjg_gaf_upload copies a GAF byte buffer to HBM (the PCIe leg); svjg_classify_resident runs the kernels
 * over the resident buffer and ADDS to the per-SV counts; svjg_classify = upload + classify_resident.
 * `base_offset` is added to the line offsets reported in hit records / input errors (for chunked files).
 * Lines end at \n, \r\n or a lone \r and the last line may be unterminated, as in Python's text mode.
 * The _file forms take the bytes [offset, offset + n_bytes) of the file itself (the `for line in aln_file` of
 * filter-alignments.py:123-126 without a host copy of the text): feeder threads pread() pieces into pinned buffers
 * that go to HBM while the next pieces are read; svjg_classify_file uses `offset` as base_offset. */
int svjg_gaf_upload(svjg_ctx *ctx, const char *gaf, uint64_t n_bytes);
int svjg_gaf_upload_file(svjg_ctx *ctx, const char *path, uint64_t offset, uint64_t n_bytes);
/* The resident text in pieces (a text larger than the caller wants to hold on the host: the 21.6 GB of BASELINE configs[3] on one
 * GPU): piece [offset, offset + n_bytes) of a text of at most capacity_bytes; the call with offset 0 sizes the buffer, the one
 * with last != 0 ends the text at offset + n_bytes and makes it resident.  Pieces go in ascending order and may be cut anywhere
 * (the text as a whole ends at a line end or without a terminator, like a file: filter-alignments.py:123-126). */
int svjg_gaf_upload_part(svjg_ctx *ctx, const char *gaf, uint64_t n_bytes, uint64_t offset, uint64_t capacity_bytes, int last);
int svjg_classify_resident(svjg_ctx *ctx, uint64_t base_offset, int want_hits);
int svjg_classify(svjg_ctx *ctx, const char *gaf, uint64_t n_bytes, uint64_t base_offset, int want_hits);
int svjg_classify_file(svjg_ctx *ctx, const char *path, uint64_t offset, uint64_t n_bytes, int want_hits);
int svjg_reset_counts(svjg_ctx *ctx);
int svjg_get_stats(svjg_ctx *ctx, svjg_stats *out);
int svjg_input_error(svjg_ctx *ctx, int *exc_class, uint64_t *line_offset);
/* Why lines took the exact string path since the last svjg_reset_counts (their sum is svjg_stats.n_deferred):
 * out[0] columns that are not twelve plain ones (blanks, signs, too few, Alen = 0, ...), [1] a 64-byte span of the line holds the
 * byte pair "d:" and the line is not decided in the main kernel (an id:f: tag, filter-alignments.py:193-196, whose value is not a
 * plain decimal, or several pairs in the line's spans), [2] a path of 4 Gbp and more (the main kernel's path sums are 32 bits wide; paths of
 * up to 216 nodes — more marks than that and the line's stripe cannot list it: [4] — stay in the main kernel), [3] a node name
 * the kernel's name table does not hold (not in the graph, a substring of another name, longer than 48 bytes, alt node without a
 * length), [4] whole stripes of 8 KB (more than 64 lines or 216 orientation marks, a line whose columns and path run beyond 8 KB or whose tail beyond them is not plain, SVJG_GRAPH_ALL_SLOW / _DOVER_LIST),
 * [5..7] reserved (0). */
int svjg_get_defer_causes(svjg_ctx *ctx, uint64_t *out8);

/* counts: out[slot*2 + 0] = ref, out[slot*2 + 1] = alt  (= len() of the two lists of
 * dict_of_informative_aln[sv_id], filter-alignments.py:163-166) */
int svjg_get_counts(svjg_ctx *ctx, uint32_t *out, uint32_t n_slots);
int svjg_set_counts(svjg_ctx *ctx, const uint32_t *in, uint32_t n_slots);   /* predict-genotype.py run stand-alone from a JSON */
/* size the count vector without a graph (stand-alone predict-genotype.py: counts come from the JSON) */
int svjg_alloc_counts(svjg_ctx *ctx, uint32_t n_slots);
/* hit records accumulated since the last svjg_reset_counts, unordered; copy at most `cap` */
int svjg_get_hits(svjg_ctx *ctx, svjg_hitrec *out, uint64_t cap, uint64_t *n);
/* byte offsets (base_offset included) of the lines set aside for the host since the last svjg_reset_counts (SVJG_EXC_ASK_HOST),
 * unordered; copy at most `cap`, *n = how many there are */
int svjg_get_host_lines(svjg_ctx *ctx, uint64_t *out, uint64_t cap, uint64_t *n);

/* ---- multi-GPU: one process per GPU, one all-reduce of the count vector over RCCL/xGMI ------------- */
int svjg_comm_unique_id(char *out128);                                      /* rank 0, then broadcast by the launcher */
int svjg_comm_init(svjg_ctx *ctx, const char *id128, int n_ranks, int rank);
int svjg_allreduce_counts(svjg_ctx *ctx);
/* ---- multi-GPU, one process (what the drop-in filter-alignments.py does with the GPUs it sees): one context per device,
 * communicators from ncclCommInitAll, the same all-reduce issued for every context; n == 1 is allowed (no collective).
 * Both all-reduce forms fail with SVJG_E_OVERFLOW when a per-SV count cannot be represented (>= 2^32). */
int svjg_comm_init_all(svjg_ctx *const *ctxs, int n);
/* What went wrong in the last failed svjg_comm_init_all (NUL terminated, at most cap bytes).  That call may run in a thread of its own beside
 * other calls on the same contexts, so it never writes a context's svjg_last_error. */
int svjg_comm_error(char *out, uint64_t cap);
/* Where the fused pass (svjg_run_begin) enqueues its all-reduce: 0 (default) on the compute stream, between this pass's kernels and
 * the next pass's; 1 on the second stream, in front of the pass's genotype kernel (bench.py measures both on a multi-GPU box). */
int svjg_comm_set_stream(svjg_ctx *ctx, int second_stream);
int svjg_allreduce_counts_all(svjg_ctx *const *ctxs, int n);

/* ---- genotypes (predict-genotype.py:216-227 gate, :281-325 likelihood) -----------------------------
 * Per VCF row r: sv_type[r] in {0 DEL, 1 INS, 2 INV, 3 BND}; slot[r] = count slot or 0xFFFFFFFF when the
 * row's sv_id is not a key of the edge table; ok[r] bit 0 = the reference's type/length gate (:216) passed,
 * bit 1 = a valid slot alone proves the sv_id is a key of the informative dict (counts loaded from a JSON).
 * Outputs: gt[r] in {0 "0/0", 1 "0/1", 2 "1/1", 3 "./."}, pl[r*3..] = the three PL integers,
 * raw[r*2..] = raw (ref, alt) counts, genotyped[r] = 1 if the row went through likelihood()
 * (i.e. counted by "Genotyped svs", :229).  Rows with genotyped = 0 print "./.:0:0,0:.,.,." (:237-239). */
int svjg_genotype(svjg_ctx *ctx, const uint8_t *sv_type, const uint32_t *slot, const uint8_t *ok,
                  uint64_t n_rows, uint32_t min_support, double err,
                  uint8_t *gt, int64_t *pl, uint32_t *raw, uint8_t *genotyped);
/* PL boundary guard.  The reference adds Decimal(math.log10(math.comb(rc1 + rc2, rc1))) to the three likelihoods
 * (predict-genotype.py:313): libm's log10 of an exact big integer, a double that need not be the correctly rounded logarithm the
 * kernel works with (double-double table of log10(i!)).  The two agree on every known answer, but a difference in the last
 * places, times ten, right next to an integer would turn a PL by one: out[r] = 1 for the rows of the last svjg_genotype /
 * svjg_genotype_view call in which one of the three -10 * (lik + comb) lies within 1e-6 of an integer.  The caller recomputes
 * those rows (a few per million) with the reference's own arithmetic: svjedi-graph_amd/svjg/genotype.py: exact_pl. */
int svjg_genotype_boundary(svjg_ctx *ctx, uint8_t *out, uint64_t n_rows);
/* The same without the copy into caller buffers: the four pointers look into the context's pinned host block (where the
 * device wrote the results) and stay valid until the next svjg_genotype / svjg_genotype_view / svjg_destroy on `ctx`. */
int svjg_genotype_view(svjg_ctx *ctx, const uint8_t *sv_type, const uint32_t *slot, const uint8_t *ok,
                       uint64_t n_rows, uint32_t min_support, double err,
                       const uint8_t **gt, const int64_t **pl, const uint32_t **raw, const uint8_t **genotyped);

/* ---- the whole pass in one call (what a fused svjedi-graph run and bench.py do per batch) -----------------------------------
 * svjg_set_rows copies the three per-row input arrays of svjg_genotype to the device once (they stay until the next
 * svjg_set_rows / svjg_destroy).  svjg_run_resident then does, for the resident text of svjg_gaf_upload and those rows:
 * svjg_reset_counts, svjg_classify_resident(base_offset, no hit records), svjg_allreduce_counts if the context has a
 * communicator, svjg_genotype — enqueued back to back with ONE host wait at the end (filter-alignments.py:123-166 and
 * predict-genotype.py:216-227, :281-325 with the counts handed over in HBM instead of through the JSON file).
 * Results land in the context's pinned host block, valid until the next svjg_run_resident / svjg_set_rows / svjg_destroy:
 * gt[r], raw[r*2..] as svjg_genotype; pl[r*3..] = the PLs as 32-bit integers; flags[r] bit 0 = genotyped, bit 1 = a PL of the
 * row does not fit 32 bits (ask svjg_genotype for the 64-bit values: it needs > 4e7 informative alignments for one SV);
 * boundary[r] as svjg_genotype_boundary. */
int svjg_set_rows(svjg_ctx *ctx, const uint8_t *sv_type, const uint32_t *slot, const uint8_t *ok, uint64_t n_rows);
int svjg_run_resident(svjg_ctx *ctx, uint64_t base_offset, uint32_t min_support, double err,
                      const uint8_t **gt, const int32_t **pl, const uint32_t **raw, const uint8_t **flags, const uint8_t **boundary);
/* The same in two halves, for a loop over batches: svjg_run_begin enqueues a pass (kernels on the context's stream, the copy of
 * its results to pinned host memory on a second stream) and returns; svjg_run_end waits for the OLDEST pass in flight and hands
 * out its results (valid until the second svjg_run_begin after it).  At most two passes are in flight, so the results of pass k
 * cross PCIe while pass k + 1 computes: begin(0); for k: begin(k + 1); end() -> results of k; ...; end(). */
int svjg_run_begin(svjg_ctx *ctx, uint64_t base_offset, uint32_t min_support, double err);
int svjg_run_end(svjg_ctx *ctx, const uint8_t **gt, const int32_t **pl, const uint32_t **raw, const uint8_t **flags, const uint8_t **boundary);

/* ---- host-side writer of <prefix>_informative_aln.json (libsvjg_host.so, no GPU involved) -----------------
 * Byte-identical to json.dumps(dict_of_informative_aln, sort_keys=True, indent=4) (filter-alignments.py:174-175)
 * from the hit records: sv_ids[slot] is the key of each count slot, `gaf` the same bytes that were classified. */
int svjg_write_informative_json(const char *path, const char *gaf, uint64_t n_bytes, const svjg_hitrec *recs,
                                uint64_t n_recs, const char *const *sv_ids, uint32_t n_slots, int n_threads);

/* Reader side for a stand-alone predict-genotype.py (json.load at predict-genotype.py:67-68, then only len() of the two
 * lists of each key, :219-226): one scan of the JSON text -> keys (unescaped UTF-8, NUL separated, file order) and
 * the two list lengths per key.  Buffers are malloc'd by the library and released with svjg_host_free. */
int svjg_count_informative_json(const char *path, char **keys_out, uint64_t *keys_len, uint64_t **counts_out, uint64_t *n_keys);
void svjg_host_free(void *p);

/* ---- native loader of the graph inputs (libsvjg_host.so, no GPU involved) ---------------------------------------
 * Fast path of svjedi-graph_amd/svjg/graph.py for the files construct-graph.py writes: <prefix>_svs_edges.json
 * (filter-alignments.py:95-98) and the alt-node S-lines of the GFA (:103-113) -> the svjg_graph tables above.
 * Returns SVJG_E_UNSUPPORTED for anything it does not recognise (escapes / non-ASCII in strings, duplicate keys, odd
 * node names, ...): the caller then uses the Python loader, which holds the semantics and raises what it raises. */
#define SVJG_E_UNSUPPORTED (-11)
typedef struct svjg_hostgraph svjg_hostgraph;
int  svjg_graph_load(const char *edges_json_path, const char *gfa_path, svjg_hostgraph **out);
const svjg_graph *svjg_graph_view(const svjg_hostgraph *g);           /* d_over = 100, flags = 0: the caller may copy and adjust */
int  svjg_graph_info(const svjg_hostgraph *g, const char **sv_ids_blob, uint64_t *sv_ids_len, uint32_t *n_hazard);   /* sv_ids: NUL-terminated, slot order */
void svjg_graph_free(svjg_hostgraph *g);

/* ---- native VCF rows (libsvjg_host.so, no GPU) ----------------------------------------------------
 * Fast path of svjedi-graph_amd/svjg/genotype.py for ordinary files: every data row's sv_id key (predict-genotype.py:118-211)
 * looked up among `n_keys` NUL-terminated keys (slots[i] = count slot of key i, NULL: slot = i; a repeated key: the last
 * wins, like json.load) -> the three input arrays of svjg_genotype; svjg_vcf_write then produces the output file
 * (predict-genotype.py:102-115, :248-271) from svjg_genotype's results.  slot_is_presence: ok = 3 instead of 1 (the
 * count table IS the JSON: svjg_genotype's gate bit 1).  SVJG_E_UNSUPPORTED for anything but plain ASCII rows with plain
 * decimal POS / END (and for every row the reference would crash on): the caller then runs the Python path. */
typedef struct svjg_vcf svjg_vcf;
int  svjg_vcf_load(const char *vcf_path, const char *keys_blob, uint64_t blob_len, const uint32_t *slots, uint32_t n_keys,
                   int slot_is_presence, svjg_vcf **out);
int  svjg_vcf_arrays(const svjg_vcf *v, const uint8_t **sv_type, const uint32_t **slot, const uint8_t **ok, uint64_t *n_rows);
int  svjg_vcf_write(const svjg_vcf *v, const char *out_path, const uint8_t *gt, const int64_t *pl, const uint32_t *raw,
                    const uint8_t *genotyped, uint64_t *n_genotyped);
void svjg_vcf_free(svjg_vcf *v);

/* ---- measurement hooks (bench.py): HIP-event time of the kernels of the last classify / genotype ---- */
int svjg_last_kernel_ms(svjg_ctx *ctx, float *classify_main_ms, float *classify_slow_ms, float *genotype_ms);
int svjg_sync(svjg_ctx *ctx);
/* what plain streams of n_bytes reach on this GPU right now (16 B per lane, non-temporal, four in flight per lane; best of three), in
 * GB/s: copy = a device-to-device copy, bytes read + bytes written per second; read = a kernel that only reads (and folds what it read
 * into one word) — the measured ceilings bench.py reports beside the 8 TB/s of the data sheet.  Either pointer may be NULL. */
int svjg_copy_rate(svjg_ctx *ctx, uint64_t n_bytes, double *copy_gb_per_s, double *read_gb_per_s);

#ifdef __cplusplus
}
#endif
#endif
