#!/usr/bin/env python3
"""measurement only (GPU box): bench.py's long_read block alone — 3 M long-read shaped lines resident, k_classify_main + the exact path a few
times — so that a profiler sees nothing else under that kernel name (rocprofv3 --pmc ... -- python3 tools/long_read_run.py [passes])."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench          # noqa: E402
import synth          # noqa: E402
from svjg import capi               # noqa: E402
from svjg.graph import Graph        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tmp = tempfile.mkdtemp(prefix="svjg_lr_")
lr_in = bench.long_read_inputs(synth, tmp, check=False)
ctx = capi.Context(0)
g = Graph.from_files(lr_in["pre"] + "_svs_edges.json", lr_in["pre"] + ".gfa")
ctx.load_graph(g)
ctx.upload(lr_in["gaf"])
ms = []
for i in range(n):
    ctx.reset_counts()
    ctx.classify_resident()
    ms.append(ctx.kernel_ms()[:2])
print("long_read: main / exact ms per launch", [tuple(round(x, 4) for x in m) for m in ms], "deferred", ctx.stats()["n_deferred"], ctx.defer_causes())
ctx.close()
