"""The CPU oracle (oracle/oracle_py.py, oracle/liboracle via oracle/oracle_c.py) against the golden
vectors produced by the reference itself (tests/golden/make_golden.py) and against the 40 known-answer
rows of the reference's own test-dir/expected_genotype.vcf."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_py as O

QUIRKS = sorted(f[:-4] for f in os.listdir(os.path.join(os.path.dirname(__file__), "golden", "quirks"))
                if f.endswith(".gaf"))


def _read_lines(path):
    # default text mode = universal newlines, exactly how the reference opens its inputs
    with open(path, encoding="utf-8") as fh:
        return fh.readlines()


@pytest.mark.parametrize("name", QUIRKS)
def test_py_filter_quirks(golden, name):
    q = f"{golden}/quirks"
    man = json.load(open(f"{q}/manifest.json"))[name]
    edges = O.load_edges(f"{q}/q_svs_edges.json")
    alt = O.load_alt_node_len(f"{q}/q.gfa")
    lines = _read_lines(f"{q}/{name}.gaf")
    if man["rc"] == 0:
        got = O.dump_informative(O.classify(lines, edges, alt))
        assert got == open(f"{q}/{name}.ref.json").read()
    else:
        with pytest.raises(Exception) as ei:
            O.classify(lines, edges, alt)
        assert type(ei.value).__name__ == man["error"]


def test_py_testdir_end_to_end(golden):
    t = f"{golden}/testdir"
    edges = O.load_edges(f"{t}/test_svs_edges.json")
    alt = O.load_alt_node_len(f"{t}/test.gfa")
    D = O.classify(_read_lines(f"{t}/test.gaf"), edges, alt)
    assert O.dump_informative(D) == open(f"{t}/ref_informative_aln.json").read()
    text, n = O.genotype_vcf(_read_lines(f"{t}/test.vcf"), D)
    assert text == open(f"{t}/ref_genotype.vcf").read()
    assert f"Genotyped svs: {n}\n" == open(f"{t}/ref_stdout.txt").read()
    # the reference's own known-answer rows (run_test.sh:41 compares non-header lines)
    exp = [l for l in open(f"{t}/expected_genotype.vcf") if not l.startswith("#")]
    assert [l for l in text.splitlines(True) if not l.startswith("#")] == exp
    assert len(exp) == 40


def test_py_likelihood_known_answers(golden):
    z = np.load(f"{golden}/lik/lik_kat.npz")
    cases, errs, txt = z["cases"], z["err"], z["dp_ad"]
    types = ["DEL", "INS", "INV", "BND"]
    gtc = {"0/0": 0, "0/1": 1, "1/1": 2, "./.": 3}
    step = 7  # pure-Python: sample the table (the C oracle test covers all rows)
    for i in range(0, len(cases), step):
        t, a, b, ms, g, p0, p1, p2 = (int(x) for x in cases[i])
        cnt = [a, b]
        gt, pl = O.likelihood(cnt, types[t], ms, float(errs[i]))
        assert (gtc[gt], [int(x) for x in pl]) == (g, [p0, p1, p2])
        assert f"{round(sum(cnt), 3)}:{cnt[0]},{cnt[1]}" == str(txt[i])


@pytest.mark.parametrize("tag,ms,err", [("ms3", 3, 5e-5), ("ms1", 1, 5e-5), ("ms0", 0, 5e-5), ("ms3_e1e-3", 3, 1e-3)])
def test_py_vcf_cases(golden, tag, ms, err):
    v = f"{golden}/vcf"
    D = json.load(open(f"{v}/cases_informative_aln.json"))
    text, n = O.genotype_vcf(_read_lines(f"{v}/cases.vcf"), D, ms, err)
    assert text == open(f"{v}/ref_{tag}.vcf").read()
    assert f"Genotyped svs: {n}\n" == open(f"{v}/ref_{tag}.stdout").read()


def test_py_vcf_crash_case(golden):
    v = f"{golden}/vcf"
    D = json.load(open(f"{v}/cases_informative_aln.json"))
    with pytest.raises(IndexError):
        O.genotype_vcf(_read_lines(f"{v}/err_no_end.vcf"), D)
