#!/usr/bin/env python3
"""measurement only: latency of the exact path for a handful of ordinary lines (whole small file through the exact path)."""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi
from svjg.graph import Graph
tmp = tempfile.mkdtemp(); pre = os.path.join(tmp, "c")
inf = synth.generate(pre, 20000, 10_000, 1, "del", 5, write_gaf=False, return_gaf=True)
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa", all_slow=True)
lines = inf["gaf"].tobytes().split(b"\n")[:-1]
ctx = capi.Context(0); ctx.load_graph(g)
for n in (1, 10, 100, 1000, 10000, 20000):
    data = np.frombuffer(b"\n".join(lines[:n]) + b"\n", dtype=np.uint8)
    ctx.upload(data)
    res = []
    for _ in range(3):
        ctx.reset_counts(); ctx.classify_resident(base_offset=0, want_hits=False); res.append(round(ctx.kernel_ms()[1], 3))
    print(n, "lines through the exact path: ms", res, flush=True)
