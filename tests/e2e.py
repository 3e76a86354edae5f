#!/usr/bin/env python3
"""End-to-end timing of the drop-in scripts on a synthetic case (file -> _informative_aln.json -> _genotype.vcf).

    python tests/e2e.py [c2|c3] [n_alignments]

Writes the synthetic inputs to a scratch directory, runs svjedi-graph_amd/filter-alignments.py and
predict-genotype.py exactly as the reference driver would (svjedi-graph.py:114, :124), and prints wall times, file sizes
and checks of both output files: against the reference's own sha256 where tests/golden/synth/<case>_full.json holds one for this
size, and — always — the genotyped VCF against the CPU oracles (C oracle counts over the whole GAF, Python oracle rows)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "svjedi-graph_amd")]


_S = {}


def _say(what):
    print(f"[e2e] {what}", file=sys.stderr, flush=True)


def _share(rng):
    lo, hi = rng
    if hi <= lo:
        return 0
    c, _, _n = _S["orc"].filter(np.asarray(_S["gaf"][lo:hi]), want_hits=False)
    return c


def check(name, pre, full_size):
    """the two output files against the oracles (this script lives under tests/: the oracle is test infrastructure)"""
    import hashlib
    import numpy as np
    from oracle import oracle_c, oracle_py
    out = {}
    gold = os.path.join(ROOT, "tests", "golden", "synth", f"{name}_full.json")
    if full_size and os.path.exists(gold):
        want = json.load(open(gold))

        def sha(path):
            h = hashlib.sha256()
            with open(path, "rb") as fh:
                for b in iter(lambda: fh.read(1 << 24), b""):
                    h.update(b)
            return h.hexdigest()
        out["json_is_the_reference_s"] = sha(pre + "_informative_aln.json") == want["sha256_json"]
        out["vcf_is_the_reference_s"] = sha(pre + "_genotype.vcf") == want["sha256_vcf"]
    t = time.time()
    orc = oracle_c.COracle(oracle_py.load_edges(pre + "_svs_edges.json"), oracle_py.load_alt_node_len(pre + ".gfa"))
    _say("oracle: graph loaded")
    # the C oracle over the whole GAF, one contiguous share of lines per forked worker and piece (the oracle keeps static scratch:
    # processes, not threads; this process never touches the GPU), a progress line per piece
    import multiprocessing as mp
    gaf = np.memmap(pre + ".gaf", dtype=np.uint8, mode="r")
    cores = min(len(os.sched_getaffinity(0)), 16)
    want = np.zeros((len(orc.sv_ids), 2), dtype=np.uint64)
    piece = 2 << 30
    a = 0
    _S.update(orc=orc, gaf=gaf)
    while a < gaf.size:
        b = min(gaf.size, a + piece)
        if b < gaf.size:
            b = a + int(np.flatnonzero(np.asarray(gaf[a:b]) == 10)[-1]) + 1
        nl = a + np.flatnonzero(np.asarray(gaf[a:b]) == 10)
        cuts = [a] + [int(nl[min(nl.size, (nl.size * (i + 1)) // cores) - 1]) + 1 for i in range(cores - 1)] + [b]
        with mp.get_context("fork").Pool(cores) as pool:
            for c in pool.map(_share, [(cuts[i], cuts[i + 1]) for i in range(cores)], chunksize=1):
                want += c
        a = b
        _say(f"oracle: {a} of {gaf.size} bytes")
    D = {sv: [["x"] * int(want[i, 0]), ["y"] * int(want[i, 1])] for i, sv in enumerate(orc.sv_ids) if want[i].sum()}
    text, n = oracle_py.genotype_vcf(open(pre + ".vcf").readlines(), D)
    out["vcf_equals_the_oracles"] = open(pre + "_genotype.vcf").read() == text
    out["oracle_check_s"] = round(time.time() - t, 1)
    return out


def main():
    import synth
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n_aln, n_sv, n_chrom, mix, seed = synth.CONFIGS[name]
    if len(sys.argv) > 2:
        n_aln = int(sys.argv[2])
    tmp = tempfile.mkdtemp(prefix="svjg_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    pre = os.path.join(tmp, "p")
    t = time.time()
    synth.generate(pre, n_aln, n_sv, n_chrom, mix, seed)
    res = {"case": name, "alignments": n_aln, "svs": n_sv, "generate_s": round(time.time() - t, 1),
           "gaf_bytes": os.path.getsize(pre + ".gaf")}
    _say(f"inputs written: {res}")
    amd = os.path.join(ROOT, "svjedi-graph_amd")
    t = time.time()
    p = subprocess.run([sys.executable, f"{amd}/filter-alignments.py", "-a", pre + ".gaf", "-g", pre + ".gfa", "-p", pre])
    res["filter_s"] = round(time.time() - t, 2); res["filter_rc"] = p.returncode
    res["json_bytes"] = os.path.getsize(pre + "_informative_aln.json") if p.returncode == 0 else None
    _say(f"filter-alignments.py: {res['filter_s']} s, rc {res['filter_rc']}, JSON {res['json_bytes']} bytes")
    t = time.time()
    p = subprocess.run([sys.executable, f"{amd}/predict-genotype.py", "-d", pre + "_informative_aln.json", "-v", pre + ".vcf",
                        "--minsupport", "3", "-o", pre + "_genotype.vcf"], capture_output=True, text=True)
    res["genotype_s"] = round(time.time() - t, 2); res["genotype_rc"] = p.returncode; res["genotype_stdout"] = p.stdout.strip()
    _say(f"predict-genotype.py: {res['genotype_s']} s, rc {res['genotype_rc']}")
    p = subprocess.run([sys.executable, f"{amd}/svjedi-graph.py", "-h"], capture_output=True)
    res["driver_help_rc"] = p.returncode
    if res["filter_rc"] == 0 and res["genotype_rc"] == 0:
        res.update(check(name, pre, n_aln == synth.CONFIGS[name][0]))
    print(json.dumps(res))
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
