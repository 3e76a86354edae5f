#!/usr/bin/env python3
"""bench.py — throughput of the hot path (classify + all-reduce + genotype) on synthetic GAF, N GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4shard] [--aln N_ALN]     (N > 1: one process, one thread per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...          (one process per GPU)

A step = one pass of the path over one batch: zero the count vector, classify every alignment of this rank's
GAF shard (text already resident in HBM), all-reduce the per-SV counts (N > 1), genotype every VCF row — one library call
(svjg_run_resident) with one host wait per pass.
value = alignments classified per second over all ranks.  One JSON line on rank 0.

Workload at N = 1: BASELINE.json configs[2] (10 M alignments x 100 k mixed SVs, the largest single-GPU
configuration); every further rank adds another 10 M alignments of the same synthetic stream on the same
graph (weak scaling; the stream is a pure function of (seed, line index), so rank r writes lines
[r*10M, (r+1)*10M)).  --workload c4shard = configs[3] / 8 per rank (12.5 M alignments x 500 k SVs).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); what a copy achieves is measured in the run (roofline.measured_copy_gbs)
SETTLE_S = 0.3               # untimed passes in front of the warm-up steps (clocks, queues)
MIN_SUSTAINED_S = 0.25       # the `sustained` block repeats the timed passes until they span this long (a sampler outside the process can then see them)
# the north_star workload (BASELINE.json configs[3]): split over the GPUs of the run, untimed for `value`
NORTH_STAR = {"aln": 100_000_000, "svs": 500_000, "chroms": 24, "mix": "mixed", "seed": 20260515 + 3, "piece": 12_500_000}

WORKLOADS = {
    # name: (alignments per rank, n_sv, n_chrom, mix, seed, description)
    "c2": (1_000_000, 10_000, 1, "del", 20260515 + 1, "configs[1]: 1 M GAF alignments x 10 k DEL SVs"),
    "c3": (10_000_000, 100_000, 4, "mixed", 20260515 + 2, "configs[2]: 10 M GAF alignments x 100 k mixed DEL/INS/INV/BND SVs"),
    "c4shard": (12_500_000, 500_000, 24, "mixed", 20260515 + 3, "configs[3]/8: 12.5 M GAF alignments x 500 k mixed SVs per GPU"),
}


def _source_digest():
    """sha256 over the sources a measurement depends on (kernels, C ABI, host libraries, this script)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "svjedi-graph_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "svjedi-graph_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "svjedi-graph_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "include", "svjg.h"), os.path.abspath(__file__)])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def write_git_head():
    """Where .git is (the build container): tools/_build/git_head = "<commit> <digest of the sources>", for the boxes that get the tree without .git.
    The commit is only ever reported together with a matching digest, so a stale file cannot name a commit the sources are not."""
    import subprocess
    r = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=5)
    if r.returncode == 0 and r.stdout.strip():
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "svjedi-graph_amd/csrc", "include", "bench.py"], capture_output=True, text=True, timeout=10).stdout.strip()
        os.makedirs(os.path.join(ROOT, "tools", "_build"), exist_ok=True)
        with open(os.path.join(ROOT, "tools", "_build", "git_head"), "w") as fh:
            fh.write(f"{r.stdout.strip()}{'+changes' if dirty else ''} {_source_digest()}\n")


def _git_head():
    """the commit this tree was taken from: .git here, or — on a GPU box, which gets no .git — what write_git_head() left, if the sources still match it"""
    import subprocess
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=5)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        commit, digest = open(os.path.join(ROOT, "tools", "_build", "git_head")).read().split()[:2]
        return commit[:12] + commit[40:] if digest == _source_digest() else f"unknown (the sources are not those of {commit[:12]})"
    except (OSError, ValueError):
        return None


def _passes(c, n, min_support, err, ms):
    """n passes on one context, two in flight: pass k + 1 is enqueued before the results of pass k are waited for, so they cross
    PCIe while the next pass computes (svjg_run_begin / svjg_run_end).  -> the last pass's results"""
    out = None
    if n <= 0:
        return out
    if os.environ.get("SVJG_BENCH_SYNC"):                      # measurement only: one pass at a time (no copy behind the next pass)
        for _ in range(n):
            out = c.run_resident(min_support, err)
            if ms is not None:
                ms.append(c.kernel_ms())
        return out
    c.run_begin(min_support, err)
    for _ in range(n - 1):
        c.run_begin(min_support, err)
        out = c.run_end()
        if ms is not None:
            ms.append(c.kernel_ms())
    out = c.run_end()
    if ms is not None:
        ms.append(c.kernel_ms())
    return out


def timed_steps(ctxs, steps, warmup, min_support=3, err=0.00005, outer_barrier=None):
    """`warmup` untimed and `steps` timed passes on every context of this process — one thread per context when there are
    several: their all-reduce keeps them in step — between two barriers.  A pass = zero counts, classify the resident shard,
    all-reduce (if the context has a communicator), genotype the resident rows, results to the host; one host wait per pass,
    and the results of a pass travel while the next one computes (_passes).
    -> (seconds for the timed passes, per context [(main, exact, genotype) kernel ms per pass], the last pass's outputs of context 0)"""
    import threading
    n = len(ctxs)
    ms = [[] for _ in range(n)]
    outs = [None] * n
    if n == 1:
        c = ctxs[0]
        _passes(c, warmup, min_support, err, None)
        c.sync()
        if outer_barrier:
            outer_barrier()
        t = time.perf_counter()
        outs[0] = _passes(c, steps, min_support, err, ms[0])
        c.sync()
        if outer_barrier:
            outer_barrier()
        return time.perf_counter() - t, ms, outs[0]
    bar = threading.Barrier(n + 1)
    errors = []

    def worker(i):
        try:
            c = ctxs[i]
            _passes(c, warmup, min_support, err, None)
            c.sync()
            bar.wait()                                           # start line
            outs[i] = _passes(c, steps, min_support, err, ms[i])
            c.sync()
            bar.wait()                                           # finish line
        except BaseException as e:                               # noqa: BLE001 (reported by the caller)
            errors.append(e)
            bar.abort()
    th = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(n)]
    for x in th:
        x.start()
    try:
        bar.wait()
        if outer_barrier:
            outer_barrier()
        t = time.perf_counter()
        bar.wait()
        if outer_barrier:
            outer_barrier()
        dt = time.perf_counter() - t
    except threading.BrokenBarrierError:
        dt = None
    for x in th:
        x.join(60)                                               # (a peer of a failed worker may sit in an all-reduce that never completes)
    if errors:
        raise errors[0]
    if dt is None or any(x.is_alive() for x in th):
        raise RuntimeError("a worker thread did not finish its passes")
    return dt, ms, outs[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)          # (a step is ~1.3 ms: 200 of them behind 20 untimed ones give the clocks time to settle)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--aln", type=int, default=0, help="override alignments per rank")
    ap.add_argument("--svs", type=int, default=0, help="override the number of SVs (experiments only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the untimed end-to-end block (files -> JSON -> VCF through the drop-in scripts)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the untimed north_star block (BASELINE configs[3] split over the GPUs of the run)")
    ap.add_argument("--no-long-read", action="store_true", help="skip the untimed long_read block (long-read shaped lines: paths long-tailed to 200 nodes, cg:Z: strings)")
    ap.add_argument("--no-hg002-shape", action="store_true", help="skip the untimed hg002_shape block (BASELINE configs[4]'s shape: whole-genome reads, most lines single-node)")
    ap.add_argument("--no-e2e-north-star", action="store_true", help="skip the untimed e2e_north_star block (configs[3] as files through the drop-in scripts; needs ~160 GB of /dev/shm)")
    ap.add_argument("--north-star-aln", type=int, default=NORTH_STAR["aln"], help="alignments of the north_star block (tests)")
    ap.add_argument("--north-star-svs", type=int, default=NORTH_STAR["svs"], help="SVs of the north_star block (tests)")
    args = ap.parse_args()

    # Two ways to N GPUs: under a launcher (torch.distributed.run: WORLD_SIZE ranks, one GPU each, RCCL communicator from a
    # unique id that travels over the launcher's process group), or — no launcher — this one process with one context and one
    # thread per GPU (ncclCommInitAll), which is also what the drop-in filter-alignments.py does with the GPUs it sees.
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_local = 1
    if world > 1:
        args.gpus = world
    elif args.gpus > 1:
        n_local = args.gpus
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")          # bootstrap + barriers only; the data-path collective is RCCL in libsvjg_hip

    import synth
    from svjg import capi, genotype, shard
    from svjg.graph import Graph
    if os.environ.get("SVJG_BENCH_CAPI"):            # tests only: a stand-in for the library (tests/standin_capi.py), so that the
        import importlib                             # launcher plumbing of this script can be run where there is no GPU
        capi = importlib.import_module(os.environ["SVJG_BENCH_CAPI"])

    if n_local > 1:
        have = capi.device_count()
        if have < n_local:
            sys.exit(f"bench.py --gpus {n_local}: needs {n_local} devices, found {have}")

    n_aln, n_sv, n_chrom, mix, seed, desc = WORKLOADS[args.workload]
    if args.aln:
        n_aln = args.aln
    if args.svs:
        n_sv = args.svs
        desc += f" [--svs {n_sv}]"
    n_total_ranks = world * n_local

    # ---- inputs (untimed) --------------------------------------------------------------------------
    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="svjg_bench_")
    pre = os.path.join(tmp, "w")
    inf = synth.generate(pre, 0, n_sv, n_chrom, mix, seed, write_gaf=False)
    gaf = synth.gaf_bytes(inf["tables"], seed, rank * n_local * n_aln, n_aln, threads=min(16, os.cpu_count() or 8))   # this process's first shard
    graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    rows = genotype.VcfRows(pre + ".vcf", graph.slot_of)
    t_setup = time.time() - t0
    cpu = cpu_rates(pre, gaf) if (not args.no_cpu_baseline and n_total_ranks == 1) else None   # (forks workers: before the GPU is touched)
    lr_in = None
    if not args.no_long_read and n_total_ranks == 1 and not os.environ.get("SVJG_BENCH_CAPI"):
        try:                                                     # its inputs, and (the checker) the C oracle over ALL of its lines on all cores — forks too
            lr_in = long_read_inputs(synth, tmp, check=not args.no_cpu_baseline)
        except Exception as e:                                   # noqa: BLE001
            lr_in = {"failed": f"{type(e).__name__}: {e}"[:300]}

    hg_in = None
    if not args.no_hg002_shape and n_total_ranks == 1 and not os.environ.get("SVJG_BENCH_CAPI"):
        try:
            hg_in = hg002_shape_inputs(synth, tmp, check=not args.no_cpu_baseline)
        except Exception as e:                                   # noqa: BLE001
            hg_in = {"failed": f"{type(e).__name__}: {e}"[:300]}

    ctxs = [capi.Context(local_rank + i) for i in range(n_local)]
    t_h2d = 0.0
    gaf_bytes_0 = int(gaf.size)
    for i, ctx in enumerate(ctxs):
        ctx.load_graph(graph)
        ctx.set_rows(rows.sv_type, rows.slot, rows.ok)
        text = gaf if i == 0 else synth.gaf_bytes(inf["tables"], seed, (rank * n_local + i) * n_aln, n_aln, threads=min(16, os.cpu_count() or 8))
        t1 = time.time()
        ctx.upload(text)
        t_h2d = max(t_h2d, time.time() - t1)
        del text
    rccl = None
    rccl_log = rccl_debug_capture(tmp) if n_total_ranks > 1 else None     # (NCCL_DEBUG must be in the environment before the communicator exists)
    t_comm = time.perf_counter()
    if world > 1:
        import dist_boot
        getattr(capi, "RcclGroup", shard.RcclGroup)(ctxs[0], world, rank, dist_boot.torch_exchange)
        rccl = {"ranks": world, "init": "ncclCommInitRank, one process per GPU", "init_s": round(time.perf_counter() - t_comm, 3)}
    elif n_local > 1:
        capi.comm_init_all(ctxs)
        rccl = {"ranks": n_local, "init": "ncclCommInitAll, one process, one thread per GPU", "init_s": round(time.perf_counter() - t_comm, 3)}
    # (init_s: what creating the communicators cost this rank, unique id's trip included — the first RCCL call of a process pays ~6 s of
    #  one-time start on the one-GPU box, profiles/r06/rccl_probe.txt; the drop-in filter-alignments.py makes them in front of its first upload,
    #  or with SVJG_COMM_OVERLAP=1 in a thread of its own beside upload + classify: svjg/filter.py: _CommInit)

    outer = dist.barrier if dist is not None else None

    def max_over_ranks(v):
        if dist is None:
            return v
        import torch
        tt = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0])

    # untimed, in front of the W warm-up steps the caller asked for: a third of a second of the same passes, so that the GPU's clocks and
    # the runtime's queues are where a running job has them (a pass is 1.3 ms: W = 5 steps are over before a cold GPU has left its idle
    # state — one driver-style run in five measured the timed region 8 % slower than the `sustained` loop right behind it)
    t_settle, n_settle = time.perf_counter(), 0
    while True:
        timed_steps(ctxs, 20, 0, outer_barrier=outer)
        n_settle += 20
        if max_over_ranks(time.perf_counter() - t_settle) >= SETTLE_S:    # (one decision for all ranks: they leave in the same round)
            break
    dt, kms, out = timed_steps(ctxs, args.steps, args.warmup, outer_barrier=outer)
    dt = max_over_ranks(dt)
    out = tuple(np.array(x) for x in out)                     # (views of the library's pinned result block: copied before anything else runs)
    # the same passes again, as many as span MIN_SUSTAINED_S (every rank computes the same number from the timed figure all of them hold)
    n_sus = max(args.steps, int(MIN_SUSTAINED_S / max(dt / args.steps, 1e-6)) + 1)
    dt_sus, kms_sus, _ = timed_steps(ctxs, n_sus, 0, outer_barrier=outer)
    dt_sus = max_over_ranks(dt_sus)
    # N > 1: the all-reduce of a pass on the compute stream (above, `value`) and on the second stream, both measured
    second = None
    if n_total_ranks > 1 and hasattr(ctxs[0], "allreduce_on_second_stream"):
        try:                                                     # (a leg beside the point of the run: its failure is reported in its place)
            for c in ctxs:
                c.allreduce_on_second_stream(True)
            dt2, kms2, _ = timed_steps(ctxs, max(args.steps, n_sus // 4), args.warmup, outer_barrier=outer)
            dt2 = max_over_ranks(dt2)
            second = {"steps": max(args.steps, n_sus // 4), "ms_per_step": dt2 / max(args.steps, n_sus // 4) * 1e3,
                      "classify_main_ms": float(np.mean([m[0] for per in kms2 for m in per]))}
        except Exception as e:                                   # noqa: BLE001
            if world > 1:                                        # (its peers sit in this leg's collectives: the launcher must end them)
                raise
            second = {"failed": f"{type(e).__name__}: {e}"[:300]}
        finally:
            for c in ctxs:
                c.allreduce_on_second_stream(False)
    main_ms = [m[0] for per in kms for m in per]
    slow_ms = [m[1] for per in kms for m in per]
    geno_ms = [m[2] for per in kms for m in per]

    ctx = ctxs[0]
    st = ctx.stats()
    counts = ctx.counts()
    gt, pl, raw, done = out
    copy_gbs = read_gbs = None
    if rank == 0 and hasattr(ctx, "copy_rate"):
        try:
            copy_gbs, read_gbs = ctx.copy_rate(1 << 31)          # 2 GB read + 2 GB written / 2 GB read, in this process, on this GPU
        except Exception as e:                                   # noqa: BLE001 (a measurement beside the point of the run)
            sys.stderr.write(f"[bench] copy rate not measured: {e}\n")
    # the untimed blocks must never cost the line its `value`: a failure in one of them is reported in its place
    ns = None
    e2e_ns_dir = e2e_ns_skip = None                              # the files of configs[3] for the e2e_north_star leg (rank 0 writes the text as it uploads it), or why there are none
    if not args.no_north_star:
        if (not args.no_e2e_north_star and not args.no_e2e and rank == 0 and n_total_ranks == 1 and not os.environ.get("SVJG_BENCH_CAPI")
                and args.north_star_aln == NORTH_STAR["aln"] and args.north_star_svs == NORTH_STAR["svs"]):
            e2e_ns_dir, e2e_ns_skip = e2e_scratch(E2E_NS_BYTES)
        try:
            ns = north_star_block(capi, synth, genotype, Graph, ctxs, rank, world, n_local, dist, args, tmp, tee_dir=e2e_ns_dir)
        except Exception as e:                                   # noqa: BLE001
            if world > 1:                                        # (a rank that leaves the block early leaves its peers in the block's collectives:
                raise                                            #  under a launcher the failure ends the run, non-zero, instead of hanging it)
            ns = {"failed": f"{type(e).__name__}: {e}"[:300]}
    lr = None
    if lr_in is not None:
        try:
            lr = long_read_block(Graph, ctx, lr_in) if "failed" not in lr_in else lr_in
        except Exception as e:                                   # noqa: BLE001
            lr = {"failed": f"{type(e).__name__}: {e}"[:300]}
        lr_in = None
    hg = None
    hg_files = None                                              # (kept for the block's end-to-end leg, which runs when this process has let go of the GPU)
    if hg_in is not None:
        try:
            hg = hg002_shape_block(Graph, ctx, hg_in) if "failed" not in hg_in else hg_in
            if "failed" not in hg_in and rank == 0 and not args.no_e2e:
                hg_files = {"pre": hg_in["pre"], "gaf": hg_in["gaf"]}
        except Exception as e:                                   # noqa: BLE001
            hg = {"failed": f"{type(e).__name__}: {e}"[:300]}
        hg_in = None

    if rank == 0:
        total_aln = n_aln * n_total_ranks
        ms_per_step = dt / args.steps * 1e3
        k_main = float(np.mean(main_ms))
        achieved = gaf_bytes_0 / (k_main * 1e-3) / 1e9
        res = {
            "metric": "GAF alignments classified/sec (classify + count all-reduce + genotype, text resident in HBM)",
            "value": total_aln * args.steps / dt,
            "unit": "alignments/s",
            "n_gpus": n_total_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int64 (classify), f64 (likelihood)", "data": "synthetic",
            "config": {"workload": desc, "alignments_per_gpu": n_aln, "svs": n_sv, "gaf_bytes_per_gpu": gaf_bytes_0,
                       "bytes_per_alignment": round(gaf_bytes_0 / n_aln, 1), "count_slots": graph.n_slots,
                       "graph_nodes": graph.n_nodes, "vcf_rows": int(len(rows.sv_type)),
                       "untimed_settle_passes_before_warmup": n_settle, "git": _git_head()},
            # (every rank genotypes ALL rows from the summed counts — 54 B a row, replicated rather than sharded: rows per second of ONE rank's kernel)
            "svs_genotyped_per_s": float(len(rows.sv_type) / (np.mean(geno_ms) * 1e-3)) if np.mean(geno_ms) > 0 else None,
            "genotyped_rows": int((done & 1).sum()),
            "kernel_ms": {"classify_main": k_main, "classify_exact_path": float(np.mean(slow_ms)), "genotype": float(np.mean(geno_ms))},
            # (two passes in flight: the genotype kernel of pass k and the transfer of its results run beside pass k + 1's classify
            #  kernel on a second stream, so only the kernels of the compute stream count here)
            "step_overhead_ms": ms_per_step - (k_main + float(np.mean(slow_ms)) + (float(np.mean(geno_ms)) if os.environ.get("SVJG_BENCH_SYNC") else 0.0)),
            "deferred_lines_per_step": st["n_deferred"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_classify_main", "algorithmic_bytes_per_launch": gaf_bytes_0},
            "setup_s": {"generate_and_tables": round(t_setup, 1), "h2d_upload": round(t_h2d, 3), "settle_passes_untimed": n_settle,
                        "pcie_inclusive_alignments_per_s": n_aln / (t_h2d + ms_per_step * 1e-3)},
            # the timed passes again, repeated until they span >= MIN_SUSTAINED_S: per-step mean over all of them
            "sustained": {"steps": n_sus, "seconds": dt_sus, "ms_per_step": dt_sus / n_sus * 1e3, "alignments_per_s": total_aln * n_sus / dt_sus,
                          "classify_main_ms": float(np.mean([m[0] for per in kms_sus for m in per]))},
        }
        if copy_gbs:
            res["roofline"]["measured_copy_gbs"] = copy_gbs      # bytes read + written per second by a plain copy kernel in this process
            res["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
            res["roofline"]["measured_read_gbs"] = read_gbs      # bytes per second of a kernel that only reads (what this read-dominated path could at best stream at)
            res["roofline"]["frac_of_measured_read"] = achieved / read_gbs if read_gbs else None
        if ns is not None:
            res["north_star"] = ns
        if lr is not None:
            res["long_read"] = lr
        if hg is not None:
            res["hg002_shape"] = hg
        if rccl:
            rccl["allreduce_stream"] = {"compute": {"ms_per_step": ms_per_step, "classify_main_ms": k_main}, "second": second,
                                        "value_is": "compute (the all-reduce between this pass's kernels and the next pass's)"}
            rccl.update(rccl_debug_summary(rccl_log))
            res["rccl"] = rccl
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes of this same command (profiles/<round>/traffic.json,
        # the newest round that has one)
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "traffic.json")))
            tr_path = cands[-1]
            tr = json.load(open(tr_path)).get(args.workload)
            if tr and tr["text_bytes"] == gaf_bytes_0 and n_total_ranks == 1:
                rel = os.path.relpath(tr_path, ROOT)
                res["roofline"]["traffic"] = tr["traffic_bytes"]
                res["roofline"]["traffic_source"] = f"{rel} (rocprofv3 FETCH_SIZE + WRITE_SIZE, gfx950 correction)"
                if tr.get("valu_wave_instructions"):
                    # what actually bounds the kernel: VALU issue (from the committed SQ_INSTS_VALU pass; a wave instruction holds
                    # its SIMD for 4 cycles, 1024 SIMDs), reported beside the HBM roofline the contract asks for
                    clk = tr.get("clock_ghz", 2.4)
                    issue_ms = tr["valu_wave_instructions"] * 4 / 1024 / (clk * 1e9) * 1e3
                    res["roofline"]["valu_issue"] = {"wave_instructions_per_launch": tr["valu_wave_instructions"], "clock_ghz": clk,
                                                     "issue_ms": round(issue_ms, 3), "frac_of_launch": round(issue_ms / k_main, 3)}
        except (OSError, ValueError, KeyError, IndexError):
            pass
        if not args.no_cpu_baseline and n_total_ranks == 1:    # reported on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(pre, graph, counts, cpu)
        if not args.no_e2e and n_total_ranks == 1 and args.workload in ("c2", "c3") and not args.aln and not args.svs:
            for c in ctxs:
                c.close()                                        # (the scripts open the GPU themselves)
            ctxs = []
            res["e2e"] = end_to_end(args.workload, pre, gaf)
        if hg_files is not None and isinstance(res.get("hg002_shape"), dict) and "failed" not in res["hg002_shape"]:
            for c in ctxs:
                c.close()
            ctxs = []
            res["hg002_shape"]["e2e"] = end_to_end_hg002(hg_files)
            hg_files = None
        if e2e_ns_dir is not None:
            for c in ctxs:
                c.close()
            ctxs = []
            res["e2e_north_star"] = end_to_end_north_star(e2e_ns_dir, ns)
        elif e2e_ns_skip is not None:
            res["e2e_north_star"] = {"skipped": e2e_ns_skip}
        print(json.dumps(res))
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def rccl_debug_capture(tmp):
    """RCCL's own account of what it set up (record algorithm / protocol / channels): NCCL_DEBUG=INFO into a file of this
    process, parsed once after the passes.  Leaves a caller's own NCCL_DEBUG settings alone."""
    if os.environ.get("NCCL_DEBUG", "").upper() in ("INFO", "TRACE") and os.environ.get("NCCL_DEBUG_FILE"):
        return os.environ["NCCL_DEBUG_FILE"]                      # (the caller logs already: read that file)
    path = os.path.join(tmp, "rccl.%h.%p.log")
    os.environ["NCCL_DEBUG"] = "INFO"                             # (over a weaker setting such as VERSION / WARN as well)
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING,ENV")   # (not COLL: that logs a line per collective call, inside the timed passes)
    os.environ["NCCL_DEBUG_FILE"] = path
    return path


def rccl_debug_summary(path):
    """-> what the log says about rings / trees, channels, transports and the all-reduce's algorithm and protocol (best effort: the wording
    differs between RCCL versions; the matching lines themselves are kept, a few of each kind)"""
    import glob
    import re
    out = {}
    try:
        files = glob.glob(re.sub(r"%[hp]", "*", path)) if path else []
        lines = []
        for f in files[:1]:
            lines = open(f, errors="replace").read().splitlines()
        if not lines:
            return {"debug_info": {"note": "no RCCL log (NCCL_DEBUG_FILE not written)"}}
        def grab(pat, n=3):
            return [re.sub(r"^.*?NCCL INFO ", "", l)[:200] for l in lines if re.search(pat, l)][:n]
        ch = [int(m.group(1)) for l in lines for m in [re.search(r"(\d+) coll channels", l)] if m]
        out["coll_channels"] = ch[0] if ch else None
        out["channels"] = grab(r"Channel \d+/\d+ *:", 2)
        out["rings_trees"] = grab(r"\bRing \d+ *:|\bTrees? \[|Connected all (rings|trees)", 4)
        out["transport"] = sorted({m.group(1) for l in lines for m in [re.search(r"via (P2P/[A-Za-z/]+|SHM[/A-Za-z]*|NET/[A-Za-z0-9]+)", l)] if m})
        algo = [m.groups() for l in lines for m in [re.search(r"[Aa]lgo(?:rithm)? *[:=]? *(\w+).*?[Pp]roto(?:col)? *[:=]? *(\w+)", l)] if m]
        out["algo_proto"] = sorted({f"{a}/{p}" for a, p in algo})[:6] or None
        out["allreduce_lines"] = grab(r"AllReduce", 3)
        out["log_lines"] = len(lines)
    except (OSError, ValueError) as e:
        out["debug"] = str(e)
    return {"debug_info": out}


def long_read_inputs(synth, tmp, check=True):
    """The long_read block's inputs and — `check`, the checker, not the thing measured — the C oracle's counts over ALL of its lines, one
    contiguous share of lines per forked worker: run before this process touches the GPU."""
    n_lines, n_sv, seed = int(os.environ.get("SVJG_LONG_READ_LINES", 3_000_000)), 20_000, 20260515 + 9
    pre = os.path.join(tmp, "long_read")
    inf = synth.generate(pre, 0, n_sv, 8, "mixed", seed, write_gaf=False, chrom_style="ucsc")
    gaf = synth.gaf_bytes(inf["tables"], seed, 0, n_lines, threads=min(16, os.cpu_count() or 8), shape="long")
    out = {"pre": pre, "gaf": gaf, "n_lines": n_lines, "n_sv": n_sv, "oracle": _oracle_counts_parallel(pre, gaf) if check else None}
    return out


def _oracle_counts_parallel(pre, gaf):
    """The checker, not the thing measured: the C oracle's counts over ALL lines of `gaf` against the graph files at `pre`, one contiguous
    share of lines per forked worker.  Forks: call before this process touches the GPU."""
    import multiprocessing as mp
    from oracle import oracle_c, oracle_py
    t = time.perf_counter()
    orc = oracle_c.COracle(oracle_py.load_edges(pre + "_svs_edges.json"), oracle_py.load_alt_node_len(pre + ".gfa"))
    cores = min(len(os.sched_getaffinity(0)), 16)
    nl = np.flatnonzero(gaf == 10)
    cuts = [0] + [int(nl[min(nl.size, (nl.size * (i + 1)) // (4 * cores)) - 1]) + 1 for i in range(4 * cores)]   # (shares of unequal cost: four a worker)
    _FORK_STATE.update(orc=orc, gaf=gaf)
    want, lines = np.zeros((len(orc.sv_ids), 2), dtype=np.uint64), 0
    try:
        with mp.get_context("fork").Pool(cores) as pool:
            for c, n in pool.imap_unordered(_oracle_shard_counts, [(cuts[i], cuts[i + 1]) for i in range(4 * cores)], chunksize=1):
                want += c
                lines += n
    finally:
        _FORK_STATE.clear()
    return {"counts": {sv: (int(want[i, 0]), int(want[i, 1])) for i, sv in enumerate(orc.sv_ids) if want[i].sum()}, "lines": lines,
            "seconds": round(time.perf_counter() - t, 1), "cores": cores}


def hg002_shape_inputs(synth, tmp, check=True):
    """The hg002_shape block's inputs (tools/synth.py: generate_hg002 — BASELINE configs[4]'s shape) and, `check`, the C oracle's counts over
    all of its lines (forks: before the GPU is touched)."""
    n_reads = int(os.environ.get("SVJG_HG002_READS", synth.HG002_READS))
    pre = os.path.join(tmp, "hg002")
    inf = synth.generate_hg002(pre, n_reads=n_reads, write_gaf=False, return_gaf=True, threads=min(16, os.cpu_count() or 8))
    gaf = inf["gaf"]
    return {"pre": pre, "gaf": gaf, "n_lines": n_reads, "n_sv": inf["n_sv"], "n_nodes": inf["n_nodes"],
            "big_nodes": int((inf["tables"]["len"][: inf["tables"]["n_ref"]] >= (1 << 25)).sum()),
            "oracle": _oracle_counts_parallel(pre, gaf) if check else None}


def hg002_shape_block(Graph, ctx, hg):
    """Untimed for `value`: BASELINE configs[4]'s SHAPE — HG002 GIAB v0.6 Tier1 (~12.8 k DEL / INS of 50 bp .. 10 kb) on the 24 GRCh37 contigs
    at their real lengths, 30x of ~20 kb reads walked from genome positions: with ~240 kb between breakpoints nine GAF lines in ten are
    SINGLE-node paths, which filter-alignments.py:133-134 skips — the regime a real whole-genome run is in and none of configs[1..3] is
    (4-5 nodes a line).  minigraph and the data are absent here; the shape is reproduced: the graph is byte for byte the one the reference's
    construct-graph.py builds from the same VCF (tests/golden/hg002shape: sha256 checked here), and the first 200 000 lines of this very
    stream went through the reference's filter (same fixture: their counts are checked here).  Classification only, kernel time by HIP
    events; `parity`: the counts of the WHOLE block against the C oracle's."""
    import hashlib
    pre, gaf, n_lines = hg["pre"], hg["gaf"], hg["n_lines"]
    graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    ctx.load_graph(graph)
    out = {}
    try:                                                         # the sample the reference itself ran (tests/golden/make_golden.py: make_hg002shape)
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "hg002shape", "hg002shape.json")))
        sha = lambda path: hashlib.sha256(open(path, "rb").read()).hexdigest()  # noqa: E731
        graph_ok = sha(pre + "_svs_edges.json") == gold["edges_json_sha256"] and sha(pre + ".gfa") == gold["gfa_elided_sha256"]
        nl = np.flatnonzero(gaf[: gold["gaf_bytes"] + 1] == 10)
        sample = gaf[: gold["gaf_bytes"]]
        sample_ok = n_lines >= gold["n_reads"] and hashlib.sha256(sample.tobytes()).hexdigest() == gold["gaf_sha256"] and nl.size >= gold["n_reads"]
        if sample_ok:
            ctx.reset_counts()
            ctx.classify(sample)
            g = ctx.counts()
            got = {graph.sv_ids[i]: [int(g[i, 0]), int(g[i, 1])] for i in range(graph.n_slots) if g[i].sum()}
            sample_ok = got == gold["counts"]
        out["pinned_by_the_reference"] = {"graph_is_construct_graph_py_s": bool(graph_ok),
                                          "first_lines_counts_equal_the_reference_s": bool(sample_ok), "lines": gold["n_reads"]}
        full = gold.get("full")                                  # the reference ran the WHOLE block too (30x): its counts are held against the last pass's below
        if full and full["n_reads"] == n_lines and full["gaf_bytes"] == int(gaf.size):
            out["pinned_by_the_reference"]["_full_counts"] = full["counts"]
    except (OSError, ValueError, KeyError) as e:
        out["pinned_by_the_reference"] = {"failed": f"{type(e).__name__}: {e}"[:200]}
    ctx.upload(gaf)
    ms = []
    for i in range(9):
        ctx.reset_counts()
        ctx.classify_resident()
        if i >= 2:
            ms.append(ctx.kernel_ms()[:2])
    main_ms, slow_ms = float(np.mean([m[0] for m in ms])), float(np.mean([m[1] for m in ms]))
    st, cause = ctx.stats(), ctx.defer_causes()
    marks = int(((gaf == ord("<")) | (gaf == ord(">"))).sum())
    out.update({"workload": f"{n_lines} whole-genome long-read lines (30x of ~20 kb reads) x {hg['n_sv']} DEL / INS on the 24 GRCh37 contigs "
                            f"({hg['n_nodes']} nodes, {hg['big_nodes']} of them >= 2^25 bp); classification only, untimed for `value`",
                "lines": n_lines, "gaf_bytes": int(gaf.size), "bytes_per_line": round(gaf.size / n_lines, 1), "path_nodes_per_line": round(marks / n_lines, 3),
                "kernel_ms": {"classify_main": main_ms, "classify_exact_path": slow_ms},
                "lines_per_s": n_lines / ((main_ms + slow_ms) * 1e-3), "gb_per_s": gaf.size / ((main_ms + slow_ms) * 1e-3) / 1e9,
                "roofline_frac": gaf.size / (main_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "deferred_lines": int(st["n_deferred"]), "deferred_by_cause": {k: int(v) for k, v in cause.items() if v}})
    if hg["oracle"] is not None:
        g = ctx.counts()
        got = {graph.sv_ids[i]: (int(g[i, 0]), int(g[i, 1])) for i in range(graph.n_slots) if g[i].sum()}
        orc = hg["oracle"]
        out["parity"] = "bit-exact" if (got == orc["counts"] and int(st["n_lines"]) == orc["lines"] == n_lines) else "MISMATCH"
        out["parity_over"] = f"all {orc['lines']} lines, {sum(a + b for a, b in orc['counts'].values())} hits; oracle/svjg_oracle.c in {orc['cores']} forked workers, {orc['seconds']} s"
    ref_counts = out.get("pinned_by_the_reference", {}).pop("_full_counts", None)
    if ref_counts is not None:                                   # every line of the block, against what the REFERENCE's filter-alignments.py counted
        g = ctx.counts()
        got = {graph.sv_ids[i]: [int(g[i, 0]), int(g[i, 1])] for i in range(graph.n_slots) if g[i].sum()}
        out["pinned_by_the_reference"]["all_lines_counts_equal_the_reference_s"] = bool(got == ref_counts and int(st["n_lines"]) == n_lines)
    return out


def long_read_block(Graph, ctx, lr_in):
    """Untimed for `value`: what the headline workload says nothing about — long-read shaped text (tools/svjg_synth.c: svjg_synth_gaf_long;
    the hand-made originals are tests/golden/realshape): sequencer read names, UCSC contig names of up to 23 bytes, paths long-tailed to 200
    nodes (3 % beyond one node pass of 64), cg:Z: strings on a third of the lines; 3 M lines (2.2 GB, the size of the headline workload's text)
    x 20 k mixed SVs on 8 contigs.  Classification only (main kernel + exact path), kernel time by HIP events: lines per second, how many
    lines took the exact path and why; `parity`: the counts of the WHOLE block against the C oracle's (long_read_inputs)."""
    pre, gaf, n_lines, n_sv = lr_in["pre"], lr_in["gaf"], lr_in["n_lines"], lr_in["n_sv"]
    graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    ctx.load_graph(graph)
    ctx.upload(gaf)
    ms = []
    for i in range(7):
        ctx.reset_counts()
        ctx.classify_resident()
        if i >= 2:
            ms.append(ctx.kernel_ms()[:2])
    main_ms, slow_ms = float(np.mean([m[0] for m in ms])), float(np.mean([m[1] for m in ms]))
    st, cause = ctx.stats(), ctx.defer_causes()
    marks = int(((gaf == ord("<")) | (gaf == ord(">"))).sum())
    out = {"workload": f"{n_lines} long-read shaped GAF lines x {n_sv} mixed SVs on 8 UCSC-named contigs; classification only, untimed for `value`",
           "lines": n_lines, "gaf_bytes": int(gaf.size), "bytes_per_line": round(gaf.size / n_lines, 1), "path_nodes_per_line": round(marks / n_lines, 2),
           "kernel_ms": {"classify_main": main_ms, "classify_exact_path": slow_ms},
           "lines_per_s": n_lines / ((main_ms + slow_ms) * 1e-3), "gb_per_s": gaf.size / ((main_ms + slow_ms) * 1e-3) / 1e9,
           "deferred_lines": int(st["n_deferred"]), "deferred_fraction": st["n_deferred"] / n_lines, "deferred_by_cause": {k: int(v) for k, v in cause.items() if v}}
    if lr_in["oracle"] is not None:                              # the last pass's counts (the whole block) against the C oracle's over all of its lines
        g = ctx.counts()
        got = {graph.sv_ids[i]: (int(g[i, 0]), int(g[i, 1])) for i in range(graph.n_slots) if g[i].sum()}
        orc = lr_in["oracle"]
        out["parity"] = "bit-exact" if (got == orc["counts"] and int(st["n_lines"]) == orc["lines"] == n_lines) else "MISMATCH"
        out["parity_over"] = f"all {orc['lines']} lines, {sum(a + b for a, b in orc['counts'].values())} hits; oracle/svjg_oracle.c in {orc['cores']} forked workers, {orc['seconds']} s"
    return out


def end_to_end_hg002(hg):
    """Untimed for `value`: the hg002_shape block as FILES through the two drop-in scripts (750 MB GAF -> _informative_aln.json -> _genotype.vcf), against
    the sha256 of what the REFERENCE's two scripts wrote for the same files in the build container (tests/golden/hg002shape: `full`), with its run times."""
    import hashlib
    import shutil
    import subprocess
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "hg002shape", "hg002shape.json")))["full"]
        if gold["gaf_bytes"] != int(hg["gaf"].size):
            return {"skipped": "the block is not at the fixture's size"}
        base = "/dev/shm" if os.path.isdir("/dev/shm") else None
        if base is None or shutil.disk_usage(base).free < 4 * int(hg["gaf"].size):
            return {"skipped": "no memory-backed scratch space for the files"}
        work = tempfile.mkdtemp(prefix="svjg_e2e_hg_", dir=base)
    except (OSError, ValueError, KeyError) as e:
        return {"skipped": str(e)}
    try:
        p = os.path.join(work, "hg")
        for ext in (".gfa", "_svs_edges.json", ".vcf"):
            shutil.copy(hg["pre"] + ext, p + ext)
        hg["gaf"].tofile(p + ".gaf")
        amd = os.path.join(ROOT, "svjedi-graph_amd")
        env = dict(os.environ)
        env.setdefault("SVJG_DEVICES", os.environ.get("LOCAL_RANK", "0"))
        sha = lambda path: hashlib.sha256(open(path, "rb").read()).hexdigest()  # noqa: E731
        t0 = time.perf_counter()
        r1 = subprocess.run([sys.executable, os.path.join(amd, "filter-alignments.py"), "-a", p + ".gaf", "-g", p + ".gfa", "-p", p], capture_output=True, text=True, env=env, timeout=E2E_TIMEOUT_S)
        t1 = time.perf_counter()
        r2 = subprocess.run([sys.executable, os.path.join(amd, "predict-genotype.py"), "-d", p + "_informative_aln.json", "-v", p + ".vcf", "--minsupport", "3", "-o", p + "_genotype.vcf"],
                            capture_output=True, text=True, env=dict(env, SVJG_NO_HANDOFF="1"), timeout=E2E_TIMEOUT_S)
        t2 = time.perf_counter()
        if r1.returncode or r2.returncode:
            return {"failed": (r1.stderr or r2.stderr)[-300:]}
        return {"what": "the block as files through the drop-in scripts (tmpfs); predict-genotype.py WITHOUT the counts hand-off (the JSON alone)",
                "filter_s": round(t1 - t0, 2), "genotype_s": round(t2 - t1, 2), "total_s": round(t2 - t0, 2), "json_bytes": os.path.getsize(p + "_informative_aln.json"),
                "sha_json_equals_the_reference_s": sha(p + "_informative_aln.json") == gold["json_sha256"],
                "sha_vcf_equals_the_reference_s": sha(p + "_genotype.vcf") == gold["vcf_sha256"] and r2.stdout.strip() == gold["genotyped"],
                "reference_s": gold["reference_seconds"]}
    except subprocess.TimeoutExpired as e:
        return {"failed": f"timeout after {E2E_TIMEOUT_S} s: {' '.join(map(str, e.cmd))[-200:]}"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def north_star_block(capi, synth, genotype, Graph, ctxs, rank, world, n_local, dist, args, tmp, tee_dir=None):
    """Untimed for `value`: BASELINE.json configs[3] — the north_star workload, 100 M alignments x 500 k SVs — split over the GPUs of
    this run (one GPU: all of it resident, 21.6 GB of text), every rank's share generated and uploaded in pieces; then whole passes
    (zero, classify, all-reduce, genotype all 500 k rows) as in the timed loop: alignments per second over all ranks, kernel times, the
    digest of the summed count vector (equal on every rank behind the all-reduce) — and, at the full size, equal to the C oracle's over
    all 100 M lines (tests/golden/synth/c4_oracle.json, written by tests/c4_oracle_counts.py --golden; tools/digest.py is the one
    spelling of the digest).  Reuses the run's contexts.  tee_dir: the graph files are generated there and this rank's text is also written to
    <tee_dir>/c4.gaf as it is uploaded (the e2e_north_star leg's inputs)."""
    import digest
    t0 = time.time()
    total, n_sv = int(args.north_star_aln), int(args.north_star_svs)
    n_ranks = world * n_local
    per = total // n_ranks
    pre = os.path.join(tee_dir, "c4") if tee_dir else os.path.join(tmp, "north_star")
    inf = synth.generate(pre, 0, n_sv, min(NORTH_STAR["chroms"], max(1, n_sv // 50)), NORTH_STAR["mix"], NORTH_STAR["seed"], write_gaf=False)
    tee = open(pre + ".gaf", "wb") if tee_dir else None

    def teed(pieces):
        for p in pieces:
            if tee is not None:
                p.tofile(tee)
            yield p
    graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    rows = genotype.VcfRows(pre + ".vcf", graph.slot_of)
    thr = min(16, os.cpu_count() or 8)
    nbytes = []
    for i, c in enumerate(ctxs):
        g = rank * n_local + i
        lo, hi = g * per, (total if g == n_ranks - 1 else (g + 1) * per)
        c.load_graph(graph)
        c.set_rows(rows.sv_type, rows.slot, rows.ok)
        pieces = (synth.gaf_bytes(inf["tables"], NORTH_STAR["seed"], a, min(NORTH_STAR["piece"], hi - a), threads=thr)
                  for a in range(lo, hi, NORTH_STAR["piece"]))
        if hasattr(c, "upload_parts"):
            nbytes.append(c.upload_parts(teed(pieces), (hi - lo) * 320 + (1 << 20)))
        else:                                                    # (a stand-in for the library: tests)
            whole = np.concatenate(list(pieces))
            c.upload(whole)
            nbytes.append(int(whole.size))
    if tee is not None:
        tee.close()
    t_setup = time.time() - t0
    outer = dist.barrier if dist is not None else None
    steps = 3
    dt, kms, out = timed_steps(ctxs, steps, 1, outer_barrier=outer)
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    digests = [digest.counts_digest(graph.sv_ids, c.counts()) for c in ctxs]
    stats = [c.stats() for c in ctxs]
    if dist is not None:
        box = [None] * world
        dist.all_gather_object(box, digests)
        digests = [d for b in box for d in b]
    main_ms = float(np.mean([m[0] for per_ctx in kms for m in per_ctx]))
    flags = np.array(out[3])
    vs_oracle = {}
    try:                                                         # the oracles' account of this workload (committed; the oracle itself does not run here)
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "synth", "c4_oracle.json")))
        n_lines_all = sum(int(x["n_lines"]) for x in stats)
        if dist is not None:
            import torch
            tt = torch.tensor([n_lines_all], dtype=torch.int64)
            dist.all_reduce(tt)
            n_lines_all = int(tt[0])
        if total == gold["lines"] and n_sv == NORTH_STAR["svs"]:
            vs_oracle = {"digest_equals_oracle": digests[0] == gold["counts_digest"], "lines_equal_oracle": n_lines_all == gold["lines"],
                         "genotyped_equals_oracle": int((flags & 1).sum()) == gold["genotyped"], "oracle": "tests/golden/synth/c4_oracle.json"}
    except (OSError, ValueError, KeyError) as e:
        vs_oracle = {"digest_equals_oracle": None, "oracle": f"not read: {e}"[:120]}
    return {**vs_oracle, "workload": f"configs[3]: {total} GAF alignments x {n_sv} mixed SVs over {n_ranks} GPU(s), text resident in HBM; untimed for `value`",
            "alignments": total, "n_gpus": n_ranks, "alignments_per_gpu": per, "gaf_bytes_per_gpu": nbytes[0], "count_slots": graph.n_slots,
            "passes": steps, "ms_per_pass": dt / steps * 1e3, "alignments_per_s": total * steps / dt,
            "kernel_ms": {"classify_main": main_ms, "classify_exact_path": float(np.mean([m[1] for per_ctx in kms for m in per_ctx])),
                          "genotype": float(np.mean([m[2] for per_ctx in kms for m in per_ctx]))},
            "roofline_frac": nbytes[0] / (main_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if main_ms > 0 else None,
            "deferred_lines_per_pass": int(stats[0]["n_deferred"]), "genotyped_rows": int((flags & 1).sum()),
            "counts_digest": digests[0][:16], "digest_equal_across_ranks": len(set(digests)) == 1, "setup_s": round(t_setup, 1)}


def end_to_end(workload, pre, gaf):
    """Untimed for `value`: the same workload through the two drop-in scripts, files on a memory-backed file system:
    GAF file -> filter-alignments.py -> _informative_aln.json -> predict-genotype.py -> _genotype.vcf, wall time per stage,
    and whether both files have the sha256 of what the reference itself wrote (tests/golden/synth/<workload>_full.json)."""
    import hashlib
    import shutil
    import subprocess
    gold = os.path.join(ROOT, "tests", "golden", "synth", f"{workload}_full.json")
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    try:
        want = json.load(open(gold))
        if base is None or shutil.disk_usage(base).free < want["json_bytes"] + 2 * int(gaf.size) + (1 << 30):
            return {"skipped": "no memory-backed scratch space for the files"}
        work = tempfile.mkdtemp(prefix="svjg_e2e_", dir=base)
    except (OSError, ValueError, KeyError) as e:
        return {"skipped": str(e)}
    try:
        p = os.path.join(work, "w")
        for ext in (".gfa", "_svs_edges.json", ".vcf"):
            shutil.copy(pre + ext, p + ext)
        gaf.tofile(p + ".gaf")
        amd = os.path.join(ROOT, "svjedi-graph_amd")

        def sha(path):
            h = hashlib.sha256()
            with open(path, "rb") as fh:
                for b in iter(lambda: fh.read(1 << 24), b""):
                    h.update(b)
            return h.hexdigest()
        env = dict(os.environ)
        env.setdefault("SVJG_DEVICES", os.environ.get("LOCAL_RANK", "0"))   # (one GPU — this rank's — unless the caller names devices: the scripts would pick for themselves)
        t0 = time.perf_counter()
        r1 = subprocess.run([sys.executable, os.path.join(amd, "filter-alignments.py"), "-a", p + ".gaf", "-g", p + ".gfa", "-p", p],
                            capture_output=True, text=True, env=env, timeout=E2E_TIMEOUT_S)
        t1 = time.perf_counter()
        r2 = subprocess.run([sys.executable, os.path.join(amd, "predict-genotype.py"), "-d", p + "_informative_aln.json", "-v", p + ".vcf",
                             "--minsupport", "3", "-o", p + "_genotype.vcf"], capture_output=True, text=True, env=env, timeout=E2E_TIMEOUT_S)
        t2 = time.perf_counter()
        if r1.returncode or r2.returncode:
            return {"failed": (r1.stderr or r2.stderr)[-300:]}
        # predict-genotype.py the way it is specified — on a JSON it knows nothing about (predict-genotype.py:67-68, :216-226): no counts
        # hand-off, the native reader over the whole file
        r3 = subprocess.run([sys.executable, os.path.join(amd, "predict-genotype.py"), "-d", p + "_informative_aln.json", "-v", p + ".vcf",
                             "--minsupport", "3", "-o", p + "_genotype_from_json.vcf"], capture_output=True, text=True, env=dict(env, SVJG_NO_HANDOFF="1"), timeout=E2E_TIMEOUT_S)
        t3 = time.perf_counter()
        if r3.returncode:
            return {"failed": "predict-genotype.py without the hand-off: " + r3.stderr[-300:]}
        ok_json = os.path.getsize(p + "_informative_aln.json") == want["json_bytes"] and sha(p + "_informative_aln.json") == want["sha256_json"]
        ok_vcf = sha(p + "_genotype.vcf") == want["sha256_vcf"] and r2.stdout == want["genotype_stdout"]
        ok_vcf_json = sha(p + "_genotype_from_json.vcf") == want["sha256_vcf"] and r3.stdout == want["genotype_stdout"]
        return {"what": "GAF file -> filter-alignments.py -> predict-genotype.py (drop-in scripts, files on tmpfs; includes process start, "
                        "HIP initialisation, graph tables, upload, JSON and VCF writing)",
                "filter_s": round(t1 - t0, 2), "genotype_s": round(t2 - t1, 2), "total_s": round(t2 - t0, 2),
                "genotype_from_json_s": round(t3 - t2, 2), "sha_vcf_from_json_ok": bool(ok_vcf_json),
                "genotype_from_json_is": "predict-genotype.py with SVJG_NO_HANDOFF=1: the counts come from the JSON file alone (native reader, chained threads)",
                "gaf_bytes": int(gaf.size), "json_bytes": os.path.getsize(p + "_informative_aln.json"),
                "sha_ok": bool(ok_json and ok_vcf and ok_vcf_json), "sha_json_ok": bool(ok_json), "sha_vcf_ok": bool(ok_vcf),
                "reference_s": {"filter": want.get("filter_s"), "genotype": want.get("genotype_s"), "where": want.get("host")}}
    except subprocess.TimeoutExpired as e:
        return {"failed": f"timeout after {E2E_TIMEOUT_S} s: {' '.join(map(str, e.cmd))[-200:]}"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


E2E_TIMEOUT_S = 600                 # no script run of an end-to-end leg may hang the bench (their files sit in tmpfs, i.e. in memory)
E2E_NS_BYTES = 160 << 30            # e2e_north_star: 21.6 GB of GAF + 117.3 GB of _informative_aln.json + the graph files, on tmpfs


def e2e_scratch(need):
    """-> (a fresh directory on /dev/shm with `need` bytes free, None) or (None, why not).  The directory is also removed when the process
    ends, however it ends short of SIGKILL (up to 140 GB of tmpfs, i.e. of host memory, must not outlive the run)."""
    import atexit
    import shutil
    import signal
    base = "/dev/shm"
    try:
        if not os.path.isdir(base):
            return None, "no /dev/shm"
        free = shutil.disk_usage(base).free
        if free < need:
            return None, f"/dev/shm has {free >> 30} GB free, the files of configs[3] need {need >> 30} GB"
        d = tempfile.mkdtemp(prefix="svjg_e2e_ns_", dir=base)
        atexit.register(shutil.rmtree, d, True)
        for sig in (signal.SIGTERM, signal.SIGHUP):              # (an outer `timeout`: ends the process through sys.exit, so that atexit runs)
            if signal.getsignal(sig) is signal.SIG_DFL:
                signal.signal(sig, lambda n, f: sys.exit(128 + n))
        return d, None
    except OSError as e:
        return None, str(e)


def end_to_end_north_star(work, ns):
    """Untimed for `value`: BASELINE configs[3] — 100 M alignments x 500 k SVs — as FILES through the two drop-in scripts (what the
    north_star's "< 60 s" is about): c4.gaf (written by the north_star block as it uploaded the text) -> filter-alignments.py ->
    c4_informative_aln.json (117 GB) -> predict-genotype.py -> c4_genotype.vcf, wall time per stage on the GPUs SVJG_DEVICES names (default:
    one).  The VCF against the sha256 of the oracles' VCF for this workload (tests/golden/synth/c4_oracle.json); the JSON at this size has no
    reference to compare with (the reference cannot run 100 M lines): its size is reported."""
    import hashlib
    import shutil
    import subprocess
    try:
        p = os.path.join(work, "c4")
        if not isinstance(ns, dict) or "failed" in ns or not os.path.exists(p + ".gaf"):
            return {"skipped": "the north_star block did not leave its files"}
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "synth", "c4_oracle.json")))
        amd = os.path.join(ROOT, "svjedi-graph_amd")
        env = dict(os.environ)
        env.setdefault("SVJG_DEVICES", os.environ.get("LOCAL_RANK", "0"))   # (one GPU unless SVJG_DEVICES names more: "all" cuts the file over every visible GPU)
        t0 = time.perf_counter()
        r1 = subprocess.run([sys.executable, os.path.join(amd, "filter-alignments.py"), "-a", p + ".gaf", "-g", p + ".gfa", "-p", p], capture_output=True, text=True, env=env, timeout=E2E_TIMEOUT_S)
        t1 = time.perf_counter()
        if r1.returncode:
            return {"failed": "filter-alignments.py: " + r1.stderr[-300:]}
        js = os.path.getsize(p + "_informative_aln.json")
        r2 = subprocess.run([sys.executable, os.path.join(amd, "predict-genotype.py"), "-d", p + "_informative_aln.json", "-v", p + ".vcf",
                             "--minsupport", "3", "-o", p + "_genotype.vcf"], capture_output=True, text=True, env=env, timeout=E2E_TIMEOUT_S)
        t2 = time.perf_counter()
        if r2.returncode:
            return {"failed": "predict-genotype.py: " + r2.stderr[-300:]}
        h = hashlib.sha256(open(p + "_genotype.vcf", "rb").read()).hexdigest()
        # the contract path of predict-genotype.py (predict-genotype.py:67-68): the 117 GB JSON read back, no counts hand-off
        r3 = subprocess.run([sys.executable, os.path.join(amd, "predict-genotype.py"), "-d", p + "_informative_aln.json", "-v", p + ".vcf",
                             "--minsupport", "3", "-o", p + "_genotype_from_json.vcf"], capture_output=True, text=True, env=dict(env, SVJG_NO_HANDOFF="1"), timeout=E2E_TIMEOUT_S)
        t3 = time.perf_counter()
        from_json = {"failed": r3.stderr[-300:]} if r3.returncode else {
            "genotype_from_json_s": round(t3 - t2, 2), "json_gb_per_s": round(js / max(t3 - t2, 1e-9) / 1e9, 2),
            "vcf_from_json_sha_equals_oracle": hashlib.sha256(open(p + "_genotype_from_json.vcf", "rb").read()).hexdigest() == gold["vcf_sha256"]}
        return {"what": "configs[3] as files on tmpfs: GAF -> filter-alignments.py -> _informative_aln.json -> predict-genotype.py -> _genotype.vcf "
                        "(drop-in scripts; includes process start, HIP initialisation, graph tables, upload, JSON and VCF writing)",
                "devices": os.environ.get("SVJG_DEVICES", "") or "one GPU (SVJG_DEVICES unset)",
                "filter_s": round(t1 - t0, 2), "genotype_s": round(t2 - t1, 2), "total_s": round(t2 - t0, 2), "under_60_s": (t2 - t0) < 60.0,
                "gaf_bytes": os.path.getsize(p + ".gaf"), "json_bytes": js,
                "vcf_sha_equals_oracle": h == gold["vcf_sha256"], "genotype_stdout": r2.stdout.strip()[-60:],
                "without_the_counts_hand_off": from_json,
                "json_check": "none at this size: the reference cannot produce it (pinned to the reference at configs[2], e2e.sha_json_ok)"}
    except (OSError, ValueError, KeyError, subprocess.TimeoutExpired) as e:
        return {"failed": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


_FORK_STATE = {}                    # what forked oracle workers inherit (the oracle's tables and the text)


def _oracle_shard_counts(rng):
    lo, hi = rng
    want, _, n = _FORK_STATE["orc"].filter(_FORK_STATE["gaf"][lo:hi], want_hits=False)
    return want, n


def _oracle_shard(rng):
    lo, hi = rng
    want, _, n = _FORK_STATE["orc"].filter(_FORK_STATE["gaf"][lo:hi], want_hits=False)
    return n


def cpu_rates(pre, gaf):
    """The CPU side of cpu_baseline, run BEFORE this process touches the GPU (the all-cores leg forks workers): the C oracle
    (restatement of the reference's per-line algorithm) on one core — the reference itself is single-threaded Python — and on
    all the cores this process may use, and the pure-Python restatement, each on a bounded sample of the same GAF.
    -> (dict for the JSON line, the oracle, the sample and its counts for the parity spot check)"""
    from oracle import oracle_c, oracle_py
    edges, alt = oracle_py.load_edges(pre + "_svs_edges.json"), oracle_py.load_alt_node_len(pre + ".gfa")
    orc = oracle_c.COracle(edges, alt)
    n_lines = 2_000_000
    nl = np.flatnonzero(gaf[: min(gaf.size, 700 * n_lines)] == 10)
    n_lines = min(n_lines, nl.size)
    sample = gaf[: int(nl[n_lines - 1]) + 1]
    t = time.perf_counter()
    want, _, got_lines = orc.filter(sample, want_hits=False)
    dt = time.perf_counter() - t
    base = {"value": got_lines / dt, "unit": "alignments/s", "cores": 1, "kind": "port",
            "sample": f"first {got_lines} alignments of the same GAF ({sample.size} bytes), oracle/svjg_oracle.c, {dt:.1f} s; "
                      f"host has {os.cpu_count()} cores"}
    try:
        import multiprocessing as mp
        cores = min(len(os.sched_getaffinity(0)), 16)            # (a one-GPU share of the host: 16 cores)
        nl_all = np.flatnonzero(gaf == 10)
        per = min(n_lines, max(1, nl_all.size // cores))
        cuts = [0] + [int(nl_all[min(nl_all.size, per * (i + 1)) - 1]) + 1 for i in range(cores)]
        _FORK_STATE.update(orc=orc, gaf=gaf)
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(_oracle_shard, [(0, 0)] * cores)               # (workers up before the clock starts)
            t = time.perf_counter()
            done = sum(pool.map(_oracle_shard, [(cuts[i], cuts[i + 1]) for i in range(cores)], chunksize=1))
            dt = time.perf_counter() - t
        base["all_cores"] = {"value": done / dt, "unit": "alignments/s", "cores": cores,
                             "sample": f"{done} alignments of the same GAF in {cores} forked workers (one contiguous share each), {dt:.1f} s"}
    except (OSError, ValueError, AttributeError) as e:
        base["all_cores"] = {"skipped": str(e)}
    try:
        sl = bytes(gaf[: int(nl[min(100_000, n_lines) - 1]) + 1]).decode("utf-8").splitlines(True)
        t = time.perf_counter()
        oracle_py.classify(sl, edges, alt)
        dt = time.perf_counter() - t
        base["python_restatement"] = {"value": len(sl) / dt, "unit": "alignments/s", "cores": 1,
                                      "sample": f"first {len(sl)} alignments, oracle/oracle_py.py, {dt:.1f} s"}
    except (OSError, ValueError) as e:
        base["python_restatement"] = {"skipped": str(e)}
    try:                                                         # the reference itself cannot travel; its rate was measured in the build container
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "synth", "c2_full.json")))
        base["reference_python"] = {"alignments_per_s": ref["alignments_per_s"], "where": ref["host"], "config": ref["config"]}
    except (OSError, ValueError, KeyError):
        pass
    return base, orc, sample, want


def cpu_baseline(pre, graph, gpu_counts, cpu):
    """cpu_rates' figures plus a parity spot check of the GPU path on the same sample and the genotype leg."""
    from oracle import oracle_py
    from svjg import capi
    base, orc, sample, want = cpu
    c2 = capi.Context(int(os.environ.get("LOCAL_RANK", "0")))
    c2.load_graph(graph)
    c2.classify(sample)
    g = c2.counts()
    c2.close()
    exp = {sv: (int(want[i, 0]), int(want[i, 1])) for i, sv in enumerate(orc.sv_ids) if want[i].sum()}
    got = {graph.sv_ids[i]: (int(g[i, 0]), int(g[i, 1])) for i in range(graph.n_slots) if g[i].sum()}
    base["parity_on_sample"] = "bit-exact" if exp == got else "MISMATCH"
    # genotype leg: pure-Python restatement on a sample of rows
    D = {graph.sv_ids[i]: [["x"] * int(gpu_counts[i, 0]), ["y"] * int(gpu_counts[i, 1])]
         for i in range(graph.n_slots) if gpu_counts[i].sum()}
    lines = open(pre + ".vcf").readlines()[:20000]
    t = time.perf_counter()
    oracle_py.genotype_vcf(lines, D)
    dt = time.perf_counter() - t
    base["genotype_svs_per_s"] = (len([l for l in lines if not l.startswith("#")])) / dt
    return base


if __name__ == "__main__":
    main()
