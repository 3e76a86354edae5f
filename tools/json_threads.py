#!/usr/bin/env python3
"""measurement only: _informative_aln.json writer throughput against the number of threads (C3-like case, /dev/shm)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi, filter as flt
from svjg.graph import Graph
n_aln = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tmp = tempfile.mkdtemp(dir="/dev/shm"); pre = os.path.join(tmp, "w")
synth.generate(pre, n_aln, 100_000, 4, "mixed", 20260517)
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
counts, recs, data = flt.classify_sharded(g, pre + ".gaf", devices=[0])
for T in (8, 16, 64):
    t = time.perf_counter()
    capi.write_informative_json("/dev/null", data, recs, g.sv_ids, n_threads=T)
    print(T, "threads, render only (/dev/null):", round(time.perf_counter() - t, 2), "s", flush=True)
for T in (4, 6, 8, 12, 16, 24, 32, 48):
    t = time.perf_counter()
    capi.write_informative_json(pre + "_o.json", data, recs, g.sv_ids, n_threads=T)
    dt = time.perf_counter() - t
    sz = os.path.getsize(pre + "_o.json")
    print(T, "threads:", round(dt, 2), "s", round(sz / dt / 1e9, 2), "GB/s", flush=True)
    os.remove(pre + "_o.json")
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
