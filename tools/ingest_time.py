#!/usr/bin/env python3
"""Measurement only: the stages between the GAF file and the hit records on the host.

    python tools/ingest_time.py [c2|c3|c4] [map|file] [n_alignments]
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "svjedi-graph_amd")]


def main():
    import synth
    from svjg import capi, filter as flt
    from svjg.graph import Graph
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n_aln, n_sv, n_chrom, mix, seed = synth.CONFIGS[name]
    if len(sys.argv) > 3:
        n_aln = int(sys.argv[3])
    tmp = tempfile.mkdtemp(prefix="svjg_ing_", dir="/dev/shm")
    pre = os.path.join(tmp, "p")
    synth.generate(pre, n_aln, n_sv, n_chrom, mix, seed)
    g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    path = pre + ".gaf"
    n = os.path.getsize(path)

    def clock(what, f, reps=3):
        best = 1e9
        for _ in range(reps):
            t = time.perf_counter()
            r = f()
            best = min(best, time.perf_counter() - t)
        print(f"{what:58s} {best * 1e3:9.1f} ms   {n / best / 1e9:6.1f} GB/s of GAF")
        return r

    t = time.perf_counter()
    ctx = capi.Context(0)
    print(f"context: {(time.perf_counter() - t) * 1e3:.0f} ms")
    t = time.perf_counter()
    ctx.load_graph(g)
    print(f"graph tables -> device: {(time.perf_counter() - t) * 1e3:.0f} ms")
    data = flt.read_gaf(path)
    lib = ctx.lib
    # cold, as the scripts meet it (first touch of the mapping / first read of the file; includes the device allocation)
    mode = sys.argv[2] if len(sys.argv) > 2 else "map"
    t = time.perf_counter()
    if mode == "map":
        ctx.upload(data)
    else:
        ctx._chk(lib.svjg_gaf_upload_file(ctx.h, os.fsencode(path), 0, n))
    print(f"cold first upload, {mode}: {(time.perf_counter() - t) * 1e3:.0f} ms")
    clock("svjg_gaf_upload, mapped file (one hipMemcpyAsync)", lambda: ctx.upload(data))
    clock("svjg_gaf_upload_file (staged pread)", lambda: ctx._chk(lib.svjg_gaf_upload_file(ctx.h, os.fsencode(path), 0, n)))
    small = np.array(data[: 60 << 20])
    t = time.perf_counter(); ctx.upload(small); dt = time.perf_counter() - t
    print(f"svjg_gaf_upload, 60 MB pageable buffer (one hipMemcpyAsync): {dt * 1e3:.1f} ms   {small.size / dt / 1e9:.1f} GB/s")
    ctx._chk(lib.svjg_gaf_upload_file(ctx.h, os.fsencode(path), 0, n))
    def cls():
        ctx.reset_counts(); ctx.classify_resident(0, True)
    clock("classify resident text (with hit records)", cls)
    recs = clock("hit records -> host", ctx.hits)
    print(f"{len(recs)} hit records, {recs.nbytes / 1e6:.0f} MB")
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
