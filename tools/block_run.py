#!/usr/bin/env python3
"""measurement only (GPU box): one of bench.py's untimed blocks alone — long_read (3 M long-read shaped lines) or hg002_shape (BASELINE configs[4]'s
shape) — resident, k_classify_main + the exact path a few times, so that a profiler sees nothing else under that kernel name
(rocprofv3 --pmc ... -- python3 tools/block_run.py long_read|hg002_shape [passes]).  Prints one JSON line shaped like bench.py's config for tools/mk_traffic.py."""
import json
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

which = sys.argv[1] if len(sys.argv) > 1 else "long_read"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tmp = tempfile.mkdtemp(prefix="svjg_blk_")
blk = bench.long_read_inputs(synth, tmp, check=False) if which == "long_read" else bench.hg002_shape_inputs(synth, tmp, check=False)
ctx = capi.Context(0)
g = Graph.from_files(blk["pre"] + "_svs_edges.json", blk["pre"] + ".gfa")
ctx.load_graph(g)
ctx.upload(blk["gaf"])
ms = []
for i in range(n):
    ctx.reset_counts()
    ctx.classify_resident()
    ms.append(ctx.kernel_ms()[:2])
main = sum(m[0] for m in ms[1:]) / max(1, len(ms) - 1)
print(json.dumps({"block": which, "config": {"gaf_bytes_per_gpu": int(blk["gaf"].size), "count_slots": g.n_slots, "lines": blk["n_lines"]},
                  "kernel_ms": {"classify_main": main, "classify_exact_path": sum(m[1] for m in ms[1:]) / max(1, len(ms) - 1)},
                  "gb_per_s": blk["gaf"].size / (main * 1e-3) / 1e9, "deferred": int(ctx.stats()["n_deferred"]), "deferred_by_cause": {k: int(v) for k, v in ctx.defer_causes().items() if v}}))
ctx.close()
