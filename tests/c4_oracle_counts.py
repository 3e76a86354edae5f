#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (run as a child process by tests/test_gpu_parity.py::test_c4_whole_hundred_million_alignments, never by the
product): the C oracle's per-SV counts over ALL alignments of BASELINE configs[3] — 8 shards of 12.5 M lines of the synthetic
stream, each cut into one contiguous share per forked worker (the oracle keeps static scratch: processes, not threads; this
process never touches the GPU).  Writes counts (sv ids in the oracle's order) to an .npz.

    python tests/c4_oracle_counts.py PREFIX OUT.npz [n_shards] [lines_per_shard]
    python tests/c4_oracle_counts.py --golden          # tests/golden/synth/c4_oracle.json: what bench.py's north_star / e2e_north_star blocks
                                                       # are checked against on the GPU box — the digest (tools/digest.py) of the C oracle's
                                                       # count vector over all 100 M lines and the sha256 of the Python oracle's VCF for it
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth                                   # noqa: E402
from oracle import oracle_c, oracle_py         # noqa: E402

_S = {}


def _share(rng):
    lo, hi = rng
    c, _, n = _S["orc"].filter(_S["gaf"][lo:hi], want_hits=False)
    return c, n


def golden():
    import hashlib
    import json
    import shutil
    import tempfile
    import time
    import digest
    work = tempfile.mkdtemp(prefix="svjg_c4gold_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        pre = os.path.join(work, "c4")
        n_aln, n_sv, n_chrom, mix, seed = synth.CONFIGS["c4"]
        inf = synth.generate(pre, 0, n_sv, n_chrom, mix, seed, write_gaf=False)
        synth.save_tables(inf["tables"], pre)
        t0 = time.time()
        sys.argv[1:] = [pre, pre + "_oracle.npz"]
        main()
        z = np.load(pre + "_oracle.npz")
        want, ids = z["counts"], [str(x) for x in z["sv_ids"]]
        D = {sv: [["x"] * int(want[i, 0]), ["y"] * int(want[i, 1])] for i, sv in enumerate(ids) if want[i].sum()}
        text, n_geno = oracle_py.genotype_vcf(open(pre + ".vcf").readlines(), D)
        out = {"config": "BASELINE configs[3]: synth.CONFIGS['c4'] (100 M alignments x 500 k mixed SVs on 24 chromosomes)",
               "lines": int(z["lines"][0]), "hits": int(want.sum()), "svs_with_counts": int((want.sum(axis=1) > 0).sum()),
               "counts_digest": digest.counts_digest(ids, want), "genotyped": int(n_geno),
               "vcf_sha256": hashlib.sha256(text.encode()).hexdigest(), "vcf_bytes": len(text.encode()),
               "by": "oracle/svjg_oracle.c (counts, all lines) + oracle/oracle_py.py (VCF) — tests/c4_oracle_counts.py --golden", "seconds": round(time.time() - t0)}
        with open(os.path.join(ROOT, "tests", "golden", "synth", "c4_oracle.json"), "w") as fh:
            json.dump(out, fh, indent=1)
        print(out)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    pre, out = sys.argv[1], sys.argv[2]
    n_shards = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    per = int(sys.argv[4]) if len(sys.argv) > 4 else 12_500_000
    n_aln, n_sv, n_chrom, mix, seed = synth.CONFIGS["c4"]
    tables = synth.load_tables(pre)
    orc = oracle_c.COracle(oracle_py.load_edges(pre + "_svs_edges.json"), oracle_py.load_alt_node_len(pre + ".gfa"))
    cores = min(len(os.sched_getaffinity(0)), 16)
    total = np.zeros((len(orc.sv_ids), 2), dtype=np.uint64)
    lines = 0
    for r in range(n_shards):
        gaf = synth.gaf_bytes(tables, seed, r * per, per, threads=cores)
        nl = np.flatnonzero(gaf == 10)
        cuts = [0] + [int(nl[min(nl.size, (nl.size * (i + 1)) // cores) - 1]) + 1 for i in range(cores)]
        _S.update(orc=orc, gaf=gaf)
        with mp.get_context("fork").Pool(cores) as pool:
            for c, n in pool.map(_share, [(cuts[i], cuts[i + 1]) for i in range(cores)], chunksize=1):
                total += c
                lines += n
        print(f"oracle: shard {r + 1} of {n_shards}, {lines} lines so far", flush=True)
    np.savez(out, counts=total, lines=np.array([lines]), sv_ids=np.array(orc.sv_ids))


if __name__ == "__main__":
    golden() if sys.argv[1:2] == ["--golden"] else main()
