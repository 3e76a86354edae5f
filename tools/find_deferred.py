#!/usr/bin/env python3
"""measurement only (GPU box): which lines of a synthetic workload take the exact path — bisection over line ranges."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi
from svjg.graph import Graph
cfg, n = sys.argv[1], int(sys.argv[2])
n_aln, n_sv, n_chrom, mix, seed = synth.CONFIGS[cfg]
tmp = tempfile.mkdtemp(dir="/dev/shm"); pre = os.path.join(tmp, "w")
inf = synth.generate(pre, 0, n_sv, n_chrom, mix, seed, write_gaf=False)
gaf = synth.gaf_bytes(inf["tables"], seed, 0, n, threads=16)
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
c = capi.Context(0); c.load_graph(g)
nl = np.flatnonzero(gaf == 10)
def ndef(lo, hi):                        # lines [lo, hi)
    a = 0 if lo == 0 else int(nl[lo - 1]) + 1
    b = int(nl[hi - 1]) + 1
    c.reset_counts(); c.classify(gaf[a:b]); return c.stats()["n_deferred"]
todo, found = [(0, nl.size)], []
while todo and len(found) < 8:
    lo, hi = todo.pop()
    d = ndef(lo, hi)
    if not d: continue
    if hi - lo == 1: found.append(lo); continue
    mid = (lo + hi) // 2
    todo += [(lo, mid), (mid, hi)]
for i in found:
    a = 0 if i == 0 else int(nl[i - 1]) + 1
    print(i, bytes(gaf[a:int(nl[i])]).decode())
