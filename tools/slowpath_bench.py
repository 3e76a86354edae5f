#!/usr/bin/env python3
"""measurement only: cost of the exact path (k_classify_slow) as a function of the number of deferred lines.
A C2-like batch (1 M alignments, 10 k DEL SVs) with N extra alignments whose paths exceed the main kernel's node cap."""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi
from svjg.graph import Graph

tmp = tempfile.mkdtemp()
pre = os.path.join(tmp, "c")
inf = synth.generate(pre, 1_000_000, 10_000, 1, "del", 5, write_gaf=False, return_gaf=True)
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa", all_slow=bool(os.environ.get("ALL_SLOW")))
names = [n for n in g.node_names if "." not in n.split(":")[-1]]
lens = {n: int(n.split(":")[1].split("-")[1]) - int(n.split(":")[1].split("-")[0]) + 1 for n in names}
base = inf["gaf"].tobytes()
lines = base.split(b"\n")[:-1]
ctx = capi.Context(0)
ctx.load_graph(g)
rng = np.random.default_rng(1)
for n_extra in [int(x) for x in (sys.argv[1:] or ["0", "1", "100", "10000"])]:
    extra = []
    for i in range(n_extra):
        NODES = int(os.environ.get("NODES", "130"))
        st = int(rng.integers(0, len(names) - NODES - 10))
        path = names[st:st + NODES]
        tlen = sum(lens[n] for n in path)
        extra.append(f"x{i}\t{tlen}\t0\t{tlen}\t+\t{''.join('>' + n for n in path)}\t{tlen}\t5\t{tlen - 7}\t{tlen}\t{tlen}\t60\ttp:A:P".encode())
    allv = lines + extra
    order = rng.permutation(len(allv)) if n_extra else np.arange(len(allv))
    data = b"\n".join(allv[i] for i in order) + b"\n"
    arr = np.frombuffer(data, dtype=np.uint8)
    ctx.upload(arr)
    res = []
    for rep in range(4):
        ctx.reset_counts()
        t = time.perf_counter()
        ctx.classify_resident(base_offset=0, want_hits=False)
        ctx.sync()
        wall = (time.perf_counter() - t) * 1e3
        res.append((round(wall, 3),) + tuple(round(x, 3) for x in ctx.kernel_ms()[:2]))
    print(n_extra, "deferred:", ctx.stats()["n_deferred"] // 4 if False else "", "wall/main/exact ms per rep:", res, flush=True)
