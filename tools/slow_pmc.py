#!/usr/bin/env python3
"""measurement only: one all-exact-path classify of a small synthetic batch (for rocprofv3 --pmc passes over k_classify_slow)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi
from svjg.graph import Graph
tmp = tempfile.mkdtemp(); pre = os.path.join(tmp, "c")
inf = synth.generate(pre, 200_000, 10_000, 1, "del", 5, write_gaf=False, return_gaf=True)
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa", all_slow=True)
ctx = capi.Context(0); ctx.load_graph(g); ctx.upload(inf["gaf"])
for _ in range(2):
    ctx.reset_counts(); ctx.classify_resident(base_offset=0, want_hits=False)
print(ctx.kernel_ms())
