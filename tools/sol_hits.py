#!/usr/bin/env python3
"""measurement only: the count updates of a bench workload in file order, one u32 per counted hit (slot * 2 + allele), for the
memory-side skeleton (tools/ubench/skeleton <workload> <reps> <file>): the skeleton then issues the main kernel's own stream of
count updates — equal counters in neighbouring lanes where a read crosses both links of one SV, neighbours on one line.  The hits come
from the C oracle (forked workers over contiguous shares of the text); the slots are the product's (svjg.graph).
    python3 tools/sol_hits.py c3|c4shard out.u32"""
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench          # noqa: E402
import synth          # noqa: E402
from oracle import oracle_c, oracle_py      # noqa: E402
from svjg.graph import Graph                # noqa: E402

_S = {}


def _share(rng):
    lo, hi = rng
    _, hits, _ = _S["orc"].filter(_S["gaf"][lo:hi], want_hits=True, hit_cap=(hi - lo) // 20 + 1024)
    return (_S["map"][hits["sv"].astype(np.int64)] * 2 + hits["allele"].astype(np.uint32)).astype(np.uint32)


def main():
    w, out = sys.argv[1], sys.argv[2]
    n_aln, n_sv, n_chrom, mix, seed, _ = bench.WORKLOADS[w]
    tmp = tempfile.mkdtemp(prefix="svjg_sol_")
    pre = os.path.join(tmp, "w")
    inf = synth.generate(pre, 0, n_sv, n_chrom, mix, seed, write_gaf=False)
    gaf = synth.gaf_bytes(inf["tables"], seed, 0, n_aln, threads=min(16, os.cpu_count() or 8))
    graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    orc = oracle_c.COracle(oracle_py.load_edges(pre + "_svs_edges.json"), oracle_py.load_alt_node_len(pre + ".gfa"))
    slot = np.array([graph.slot_of[s] for s in orc.sv_ids], dtype=np.uint32)
    if len(sys.argv) > 3 and sys.argv[3] == "genome-order":      # experiment: count slots numbered along the genome instead of by the ids' string order
        import re

        def where(sv):
            c, rest = sv.split(":", 1)
            m = re.match(r"[A-Z]+-(\d+)", rest) or re.search(r"(\d+)$", rest)
            return (c, int(m.group(1)))
        order = sorted(range(len(graph.sv_ids)), key=lambda i: where(graph.sv_ids[i]))
        perm = np.empty(len(order), dtype=np.uint32)
        perm[np.array(order)] = np.arange(len(order), dtype=np.uint32)
        slot = perm[slot]
    _S.update(orc=orc, gaf=gaf, map=slot)
    cores = min(len(os.sched_getaffinity(0)), 16)
    nl = np.flatnonzero(gaf == 10)
    per = nl.size // cores
    cuts = [0] + [int(nl[(per * (i + 1) if i + 1 < cores else nl.size) - 1]) + 1 for i in range(cores)]
    with mp.get_context("fork").Pool(cores) as pool:
        parts = pool.map(_share, [(cuts[i], cuts[i + 1]) for i in range(cores)], chunksize=1)
    hv = np.concatenate(parts)
    hv.tofile(out)
    print(f"{w}: {hv.size} count updates of {nl.size} alignments -> {out}; equal to the lane below: {float((hv[1:] == hv[:-1]).mean()):.3f}, "
          f"on the 64-byte line of the lane below: {float(((hv[1:] >> 4) == (hv[:-1] >> 4)).mean()):.3f}")


if __name__ == "__main__":
    main()
