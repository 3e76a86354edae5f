#!/usr/bin/env python3
"""measurement only: exact-path kernels under rocprofv3 --pmc: N lines with long paths through a C2-like batch."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi
from svjg.graph import Graph
NODES = int(os.environ.get("NODES", "130")); N = int(os.environ.get("N", "1"))
tmp = tempfile.mkdtemp(); pre = os.path.join(tmp, "c")
inf = synth.generate(pre, 20000, 10_000, 1, "del", 5, write_gaf=False, return_gaf=True)
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
names = [n for n in g.node_names if "." not in n.split(":")[-1]]
lens = {n: int(n.split(":")[1].split("-")[1]) - int(n.split(":")[1].split("-")[0]) + 1 for n in names}
extra = []
for i in range(N):
    path = names[100 + i:100 + i + NODES]; tlen = sum(lens[n] for n in path)
    extra.append(f"x{i}\t{tlen}\t0\t{tlen}\t+\t{''.join('>' + n for n in path)}\t{tlen}\t5\t{tlen - 7}\t{tlen}\t{tlen}\t60\ttp:A:P\n".encode())
data = inf["gaf"].tobytes() + b"".join(extra)
ctx = capi.Context(0); ctx.load_graph(g); ctx.upload(np.frombuffer(data, dtype=np.uint8))
for _ in range(2):
    ctx.reset_counts(); ctx.classify_resident(base_offset=0, want_hits=False)
print(ctx.kernel_ms(), ctx.stats())
