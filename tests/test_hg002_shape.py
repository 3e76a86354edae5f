"""golden/hg002shape (BASELINE configs[4]'s shape; graph built by the reference's construct-graph.py, 200 000 read lines through the
reference's filter + genotyper: tests/golden/make_golden.py: make_hg002shape) against the generator and both oracles, on the CPU.
The HIP path takes the same fixture in tests/test_gpu_parity.py::test_hg002_shape."""
import hashlib
import json
import os

import numpy as np

from oracle import oracle_c as OC
from oracle import oracle_py as O


def test_generator_and_oracles_reproduce_the_reference(tmp_path, golden):
    import synth
    want = json.load(open(f"{golden}/hg002shape/hg002shape.json"))
    pre = str(tmp_path / "hg")
    inf = synth.generate_hg002(pre, n_reads=want["n_reads"], seed=want["seed"])

    def sha(path):
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    # the regenerated graph IS the one construct-graph.py built (edges JSON byte for byte, GFA with the reference sequences elided)
    assert sha(pre + "_svs_edges.json") == want["edges_json_sha256"] and sha(pre + ".gfa") == want["gfa_elided_sha256"]
    assert sha(pre + ".vcf") == want["vcf_in_sha256"] and sha(pre + ".gaf") == want["gaf_sha256"]
    assert inf["n_sv"] == want["n_sv"] == 12800 and inf["n_nodes"] == want["n_nodes"]
    lens = inf["tables"]["len"][: inf["tables"]["n_ref"]]
    assert int((lens >= (1 << 25)).sum()) >= 1                      # a whole-genome graph has nodes the r05 kernel could not keep
    assert want["single_node_lines"] > 0.9 * want["n_reads"]        # the regime: most lines cross no breakpoint
    edges, alt = O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa")
    gaf = np.fromfile(pre + ".gaf", dtype=np.uint8)
    orc = OC.COracle(edges, alt)
    cnt, _, n_lines = orc.filter(gaf, want_hits=False)
    assert n_lines == want["n_reads"]
    assert {sv: [int(cnt[i, 0]), int(cnt[i, 1])] for i, sv in enumerate(orc.sv_ids) if cnt[i].sum()} == want["counts"]
    D = O.classify(open(pre + ".gaf").readlines(), edges, alt)
    js = O.dump_informative(D)
    assert hashlib.sha256(js.encode()).hexdigest() == want["json_sha256"]
    text, n = O.genotype_vcf(open(pre + ".vcf").readlines(), D)
    assert f"Genotyped svs: {n}" == want["genotyped"] and hashlib.sha256(text.encode()).hexdigest() == want["vcf_sha256"]


def test_the_whole_block_against_the_reference(tmp_path, golden):
    """the bench block itself — all 4.65 M lines (30x) — went through the reference's filter and genotyper (fixture: `full`): the regenerated text
    has its sha256 and the C oracle counts what the reference counted, for every SV"""
    import synth
    want = json.load(open(f"{golden}/hg002shape/hg002shape.json"))
    full = want["full"]
    pre = str(tmp_path / "hg")
    inf = synth.generate_hg002(pre, n_reads=full["n_reads"], seed=want["seed"], write_gaf=False, return_gaf=True)
    gaf = inf["gaf"]
    assert int(gaf.size) == full["gaf_bytes"] and hashlib.sha256(gaf.tobytes()).hexdigest() == full["gaf_sha256"]
    orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
    cnt, _, n_lines = orc.filter(gaf, want_hits=False)
    assert n_lines == full["n_reads"]
    assert {sv: [int(cnt[i, 0]), int(cnt[i, 1])] for i, sv in enumerate(orc.sv_ids) if cnt[i].sum()} == full["counts"]
    assert len(full["counts"]) > 12000 and full["genotyped"].startswith("Genotyped svs: ")
