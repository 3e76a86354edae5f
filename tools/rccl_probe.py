#!/usr/bin/env python3
"""measurement only (GPU box): what RCCL logs about a communicator of this process's GPUs — one rank per visible GPU, ncclCommInitAll, or a
one-rank communicator on a one-GPU box — and what bench.py's parser (rccl_debug_summary) makes of it.  Prints the raw INFO lines too."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench          # noqa: E402

tmp = tempfile.mkdtemp()
log = bench.rccl_debug_capture(tmp)
import numpy as np    # noqa: E402
import synth          # noqa: E402
from svjg import capi, genotype, shard      # noqa: E402
from svjg.graph import Graph                # noqa: E402

pre = os.path.join(tmp, "w")
inf = synth.generate(pre, 20000, 500, 2, "mixed", 3, write_gaf=False, return_gaf=True)
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
rows = genotype.VcfRows(pre + ".vcf", g.slot_of)
n = capi.device_count()
ctxs = [capi.Context(i) for i in range(n)]
for c in ctxs:
    c.load_graph(g); c.set_rows(rows.sv_type, rows.slot, rows.ok); c.upload(inf["gaf"])
import time           # noqa: E402


def probe_init_all():
    """r06: the call the drop-in filter-alignments.py makes (svjg_comm_init_all = ncclCommInitAll over the process's devices), straight on librccl for
    the devices of this box — on a one-GPU box a communicator of one rank, which the library itself never creates (n == 1: no collective)"""
    import ctypes
    rccl = ctypes.CDLL("/opt/rocm/lib/librccl.so")
    comms = (ctypes.c_void_p * n)()
    devs = (ctypes.c_int * n)(*range(n))
    rccl.ncclCommInitAll.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    for rep in range(2):
        t_all = time.perf_counter()
        rc = rccl.ncclCommInitAll(comms, n, devs)
        print(f"ncclCommInitAll({n} device(s)) call {rep + 1}: rc {rc}, {time.perf_counter() - t_all:.3f} s", flush=True)
        for cm in comms:
            rccl.ncclCommDestroy(ctypes.c_void_p(cm))


# (whichever communicator a process creates FIRST pays RCCL's one-time start: `python tools/rccl_probe.py all-first` turns the order round)
if "all-first" in sys.argv:
    probe_init_all()
t_init = time.perf_counter()
if n > 1:
    capi.comm_init_all(ctxs)
else:
    shard.RcclGroup(ctxs[0], 1, 0, lambda uid: uid)
print(f"init_s: {time.perf_counter() - t_init:.3f} (communicator of {n} rank(s): {'ncclCommInitAll' if n > 1 else 'ncclCommInitRank'})", flush=True)
if "all-first" not in sys.argv:
    probe_init_all()
dt, ms, out = bench.timed_steps(ctxs, 3, 1)
print("passes ok:", dt, [m[-1] for m in ms])
import glob, re
for f in glob.glob(re.sub(r"%[hp]", "*", log)):
    print("----", f)
    print(open(f, errors="replace").read()[:6000])
print(bench.rccl_debug_summary(log))
