"""tools/synth.py lays the graph out exactly like the reference's construct-graph.py.
Runs only where /root/reference exists (the build container); skipped on the GPU box."""
import os
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference/construct-graph.py"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference not present on this machine")
@pytest.mark.parametrize("n_sv,n_chrom,mix,seed", [(300, 3, "mixed", 7), (200, 1, "del", 8)])
def test_graph_matches_construct_graph(tmp_path, n_sv, n_chrom, mix, seed):
    import synth
    pre = str(tmp_path / "s")
    inf = synth.generate(pre, 100, n_sv, n_chrom, mix, seed)
    rng = np.random.default_rng(1)
    with open(tmp_path / "ref.fa", "w") as fh:
        for c, l in zip(inf["chroms"], inf["chrom_len"]):
            s = "".join(rng.choice(list("ACGT"), size=l))
            fh.write(f">{c}\n" + "\n".join(s[i:i + 80] for i in range(0, l, 80)) + "\n")
    os.makedirs(tmp_path / "r")
    subprocess.run([sys.executable, REF, "-v", pre + ".vcf", "-r", str(tmp_path / "ref.fa"),
                    "-o", str(tmp_path / "r" / "s.gfa")], check=True, capture_output=True)
    assert open(pre + "_svs_edges.json").read() == open(tmp_path / "r" / "s_svs_edges.json").read()

    def norm(path):
        out = []
        for ln in open(path):
            if ln.startswith("S"):
                c = ln.rstrip("\n").split("\t")
                if "." not in c[1].split(":")[-1]:
                    ln = f"S\t{c[1]}\t*\n"
            out.append(ln)
        return out

    assert norm(pre + ".gfa") == norm(tmp_path / "r" / "s.gfa")
