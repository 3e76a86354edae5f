"""Native VCF rows (libsvjg_host: svjg_vcf_load / svjg_vcf_write) against the Python rows of svjg/genotype.py, which hold
the semantics of predict-genotype.py:102-271: same arrays, same output bytes; irregular files are declined."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")]
GOLD = os.path.join(ROOT, "tests", "golden")


def _both(vcf, slot_of, presence, tmp_path, seed=1):
    from svjg import capi, genotype as G
    py = G.VcfRows(vcf, slot_of if isinstance(slot_of, dict) else {k: i for i, k in enumerate(slot_of)}, presence)
    nat = G.open_rows(vcf, slot_of, presence)
    assert isinstance(nat, capi.NativeVcfRows), "the native reader declined an ordinary file"
    assert np.array_equal(py.sv_type, nat.sv_type) and np.array_equal(py.slot, nat.slot) and np.array_equal(py.ok, nat.ok)
    n = len(py.sv_type)
    rng = np.random.default_rng(seed)
    gt = rng.integers(0, 4, n).astype(np.uint8)
    pl = rng.integers(-10 ** 12, 10 ** 12, (n, 3)).astype(np.int64)
    raw = rng.integers(0, 40, (n, 2)).astype(np.uint32)
    raw[rng.integers(0, 5, n) == 0] = 0
    if n:
        raw[0] = (4294967295, 4294967295)
    done = (rng.integers(0, 3, n) > 0).astype(np.uint8)
    a, b = str(tmp_path / "py.vcf"), str(tmp_path / "nat.vcf")
    n_py = G.write_vcf(a, py, gt, pl, raw, done)
    n_nat = nat.write(b, gt, pl, raw, done)
    nat.close()
    assert n_py == n_nat == int(done.sum())
    assert open(a, "rb").read() == open(b, "rb").read()
    return n


def _keys_of(vcf):
    """every key the Python rows would look up (so that slots are exercised), in file order"""
    from svjg import genotype as G
    keys, seen = [], {}
    for line in open(vcf):
        if line.startswith("#"):
            continue
        c = line.rstrip("\n").split("\t")
        keys.append(G.row_key(c[0], c[1], c[4], c[7], seen)[1])
    return keys


@pytest.mark.parametrize("presence", [False, True])
def test_reference_fixtures(tmp_path, presence):
    for vcf in (os.path.join(GOLD, "vcf", "cases.vcf"), os.path.join(GOLD, "testdir", "test.vcf")):
        if not os.path.exists(vcf):
            continue
        keys = _keys_of(vcf)
        slot_of = {k: (7 * i) % 1000 for i, k in enumerate(keys[::2])}          # half of the rows have a slot
        assert _both(vcf, slot_of, presence, tmp_path) > 0
        assert _both(vcf, keys[::3] + keys[:5], presence, tmp_path) > 0          # keys as a list, with repeats: the last wins


def test_synthetic_mix(tmp_path):
    import synth
    pre = str(tmp_path / "s")
    synth.generate(pre, 100, 3000, 5, "mixed", 99)
    from svjg.graph import Graph
    g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    assert _both(pre + ".vcf", g.slot_of, False, tmp_path) == 3000
    assert _both(pre + ".vcf", list(g.sv_ids), True, tmp_path) == 3000


HEAD = "##fileformat=VCFv4.2\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"x\">\n##contig=<ID=1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"


def test_layouts(tmp_path):
    """column counts, SVTYPE / END first, middle and last in INFO, the four BND forms, no trailing newline, odd types"""
    rows = [
        "1\t100\t.\tN\t<DEL>\t.\t.\tSVTYPE=DEL;END=400",
        "1\t100\t.\tN\t<DEL>\t.\t.\tEND=400;SVTYPE=DEL",
        "1\t100\t.\tN\t<DEL>\t.\t.\tAC=1;END=400;SVTYPE=DEL;AF=2\tGT\t0/1",
        "1\t100\t.\tN\t<DEL>\t.\t.\tAC=1;SVTYPE=DEL;END=120\tGT",
        "1\t500\t.\tN\t" + "A" * 80 + "\t.\t.\tSVTYPE=INS",
        "2\t500\t.\tN\t" + "A" * 60 + "\t.\t.\tSVTYPE=INS;END=500",
        "2\t500\t.\tN\t<INS>\t.\t.\tSVTYPE=INS",
        "1\t700\t.\tN\t<INV>\t.\t.\tSVTYPE=INV;END=900",
        "1\t800\t.\tN\tN[2:300[\t.\t.\tSVTYPE=BND",
        "1\t800\t.\tN\t[2:300[N\t.\t.\tSVTYPE=BND",
        "1\t800\t.\tN\tN]2:300]\t.\t.\tSVTYPE=BND",
        "1\t800\t.\tN\t]2:300]N\t.\t.\tSVTYPE=BND;X=1",
        "1\t800\t.\tN\tNNN\t.\t.\tSVTYPE=BND",
        "1\t900\t.\tN\t<DUP>\t.\t.\tSVTYPE=DUP;END=2000",
        "1\t900\t.\tN\t<X>\t.\t.\tFOO=1;END=2000",
        "1\t0100\t.\tN\t<DEL>\t.\t.\tXSVTYPE=a;SVTYPE=DEL;END=400",
        "1\t100\t.\tN\t<DEL>\t.\t.\tSVTYPE=DEL;END=400;SVTYPE=DEL",
        "1\t100\t.\tN\t<DEL>\t.\t.\tSVTYPE=DEL;XEND=7;END=400",
    ]
    vcf = str(tmp_path / "l.vcf")
    open(vcf, "w").write(HEAD + "\n".join(rows))                                 # last line unterminated
    keys = _keys_of(vcf)
    assert _both(vcf, {k: i for i, k in enumerate(keys)}, True, tmp_path) == len(rows)
    open(vcf, "w").write(HEAD + "\n".join(rows) + "\n")
    assert _both(vcf, keys, False, tmp_path) == len(rows)
    open(vcf, "w").write(HEAD)
    assert _both(vcf, keys, False, tmp_path) == 0


def test_irregular_files_are_left_to_python(tmp_path):
    from svjg import capi, genotype as G
    ok = "1\t100\t.\tN\t<DEL>\t.\t.\tSVTYPE=DEL;END=400\n"
    cases = {
        "crlf": HEAD + ok.replace("\n", "\r\n"),
        "utf8": HEAD + ok.replace("<DEL>", "<DÉL>"),
        "seven_columns": HEAD + "1\t100\t.\tN\t<DEL>\t.\t.\n",
        "empty_line": HEAD + ok + "\n" + ok,
        "no_end": HEAD + "1\t100\t.\tN\t<DEL>\t.\t.\tSVTYPE=DEL\n",
        "svtype_without_value": HEAD + "1\t100\t.\tN\t<DEL>\t.\t.\tSVTYPE;END=400\n",
        "signed_pos": HEAD + "1\t+100\t.\tN\t<DEL>\t.\t.\tSVTYPE=DEL;END=400\n",
        "blank_in_end": HEAD + "1\t100\t.\tN\t<DEL>\t.\t.\tSVTYPE=DEL;END= 400\n",
        "underscore": HEAD + "1\t1_00\t.\tN\t<DEL>\t.\t.\tSVTYPE=DEL;END=400\n",
        "bnd_one_part": HEAD + "1\t800\t.\tN\t[2:300[\t.\t.\tSVTYPE=BND\n",
        "other_hash_line": HEAD + "#x\t1\t2\t3\t4\t5\t6\tSVTYPE=DEL;END=400\n",
    }
    for name, text in cases.items():
        p = str(tmp_path / f"{name}.vcf")
        open(p, "w", encoding="utf-8").write(text)
        assert capi.vcf_load_native(p, ["1:DEL-100-400"]) is None, name
        rows = None
        try:
            rows = G.open_rows(p, ["1:DEL-100-400"])                              # falls back; raises only what the Python rows raise
        except (ValueError, IndexError):
            pass
        assert rows is None or isinstance(rows, G.VcfRows), name
    assert capi.vcf_load_native(str(tmp_path / "missing.vcf"), []) is None
    p = str(tmp_path / "ok.vcf")
    open(p, "w").write(HEAD + ok)
    r = capi.vcf_load_native(p, ["1:DEL-100-400"], None, True)
    assert r is not None and r.slot.tolist() == [0] and r.ok.tolist() == [3] and r.sv_type.tolist() == [0]
    os.environ["SVJG_PY_VCF"] = "1"
    try:
        assert isinstance(G.open_rows(p, ["1:DEL-100-400"]), G.VcfRows)
    finally:
        del os.environ["SVJG_PY_VCF"]


def test_exact_pl_is_the_reference_arithmetic(golden):
    """svjg.genotype.exact_pl (what recomputes the rows the kernel flags as lying next to a PL's integer boundary) on the
    reference's known answers: the near-boundary and deep-sample cases of lik_boundary.npz and a slice of the main table."""
    import numpy as np
    from svjg import genotype
    for name, step in (("lik_boundary.npz", 1), ("lik_kat.npz", 37)):
        z = np.load(f"{golden}/lik/{name}")
        cases, errs = z["cases"][::step], z["err"][::step]
        for c, e in zip(cases.tolist(), errs.tolist()):
            if c[1] + c[2] > 20000:
                continue                                           # (seconds each in math.comb; the GPU test covers them)
            assert genotype.exact_pl(c[0], c[1], c[2], e) == c[5:8], c
