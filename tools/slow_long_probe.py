#!/usr/bin/env python3
"""measurement only (GPU box): what ONE long line costs the exact path — N copies of a line of ~200 nodes and > 8 KB (longer than a stripe:
the main kernel defers it whole) among ordinary lines, N = 1, 8, 64, 512: latency of a line against throughput of the kernel."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi
from svjg.graph import Graph

tmp = tempfile.mkdtemp(); pre = os.path.join(tmp, "w")
seed = 20260515 + 9
inf = synth.generate(pre, 0, 20000, 8, "mixed", seed, write_gaf=False, chrom_style="ucsc")
gaf = synth.gaf_bytes(inf["tables"], seed, 0, 300000, threads=16, shape="long")
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa", all_slow=True)     # (r04: a plain tail no longer sends a line to the exact path: every line is sent)
lines = bytes(gaf).split(b"\n")[:-1]
longest = max(lines, key=len)
k = longest.split(b"\t")[5].count(b">") + longest.split(b"\t")[5].count(b"<")
pad = b"".join(l + b"\n" for l in lines[:20] if len(l) < 2000)
ctx = capi.Context(0); ctx.load_graph(g)
print(f"the line: {len(longest)} bytes, {k} nodes")
for n in (1, 8, 64, 512):
    big = longest + b"x" * max(0, 8300 - len(longest))            # (a tag that runs past the stripe)
    data = np.frombuffer(pad + (big + b"\n") * n + pad, dtype=np.uint8)
    ctx.upload(data)
    res = []
    for rep in range(4):
        ctx.reset_counts(); ctx.classify_resident()
        res.append(tuple(round(x, 3) for x in ctx.kernel_ms()[:2]))
    print(n, "lines:", ctx.stats()["n_deferred"], "deferred; main / exact ms:", res[1:], flush=True)
