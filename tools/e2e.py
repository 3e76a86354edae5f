#!/usr/bin/env python3
"""End-to-end timing of the drop-in scripts on a synthetic case (file -> _informative_aln.json -> _genotype.vcf).

    python tools/e2e.py [c2|c3] [n_alignments]

Writes the synthetic inputs to a scratch directory, runs svjedi-graph_amd/filter-alignments.py and
predict-genotype.py exactly as the reference driver would (svjedi-graph.py:114, :124), and prints wall times, file sizes
and a check of the genotyped VCF against the CPU oracle on the first rows."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "svjedi-graph_amd")]


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
    amd = os.path.join(ROOT, "svjedi-graph_amd")
    t = time.time()
    p = subprocess.run([sys.executable, f"{amd}/filter-alignments.py", "-a", pre + ".gaf", "-g", pre + ".gfa", "-p", pre])
    res["filter_s"] = round(time.time() - t, 2); res["filter_rc"] = p.returncode
    res["json_bytes"] = os.path.getsize(pre + "_informative_aln.json") if p.returncode == 0 else None
    t = time.time()
    p = subprocess.run([sys.executable, f"{amd}/predict-genotype.py", "-d", pre + "_informative_aln.json", "-v", pre + ".vcf",
                        "--minsupport", "3", "-o", pre + "_genotype.vcf"], capture_output=True, text=True)
    res["genotype_s"] = round(time.time() - t, 2); res["genotype_rc"] = p.returncode; res["genotype_stdout"] = p.stdout.strip()
    t = time.time()
    p = subprocess.run([sys.executable, f"{amd}/svjedi-graph.py", "-h"], capture_output=True)
    res["driver_help_rc"] = p.returncode
    print(json.dumps(res))
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
