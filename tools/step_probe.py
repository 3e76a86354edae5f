#!/usr/bin/env python3
"""Measurement only (GPU box): host wall time of the three calls of a bench step, step by step, to see where a step's time goes
beyond its kernels (C2-sized input; `python tools/step_probe.py [n_steps]`)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "svjedi-graph_amd"))
import numpy as np     # noqa: E402
import synth           # noqa: E402
from svjg import capi, genotype   # noqa: E402
from svjg.graph import Graph      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if os.environ.get("PROBE_CPUS"):                     # e.g. 0-63: run on these CPUs only (set before anything touches the GPU)
    lo, hi = os.environ["PROBE_CPUS"].split("-")
    os.sched_setaffinity(0, range(int(lo), int(hi) + 1))
pre = os.path.join(tempfile.mkdtemp(prefix="svjg_probe_"), "w")
inf = synth.generate(pre, 0, 100000, 24, "mixed", 3, write_gaf=False)
gaf = synth.gaf_bytes(inf["tables"], 3, 0, 10000000, threads=16)
graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
rows = genotype.VcfRows(pre + ".vcf", graph.slot_of)
ctx = capi.Context(0)
ctx.load_graph(graph)
ctx.upload(gaf)
print("sched_getaffinity:", len(os.sched_getaffinity(0)), "cpus; on cpu", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
import glob
for f in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
    print(f, open(f).read().strip(), open(os.path.join(os.path.dirname(f), "vendor")).read().strip())
for f in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(f.split("/")[-2], open(f).read().strip())
print("affinity:", sorted(os.sched_getaffinity(0))[:40], "model:", [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1])
acc = []
WARM = int(os.environ.get('PROBE_WARM', '3'))
for i in range(n + WARM):
    t0 = time.perf_counter(); ctx.reset_counts()
    t1 = time.perf_counter(); ctx.classify_resident(base_offset=0, want_hits=False)
    t2 = time.perf_counter(); ctx.genotype(rows.sv_type, rows.slot, rows.ok, 3, 0.00005, reuse_outputs=True)
    t3 = time.perf_counter()
    if i >= WARM:
        acc.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
a = np.array(acc)
print("ms per call  reset / classify / genotype: mean %s  min %s  max %s" % (a.mean(0).round(3), a.min(0).round(3), a.max(0).round(3)))
print("genotype call, step by step:", " ".join("%.2f" % x for x in a[:, 2]))
print("kernel ms (main, exact, genotype):", [round(x, 3) for x in ctx.kernel_ms()])
