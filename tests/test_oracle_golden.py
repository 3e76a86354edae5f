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


UNICODE = sorted(f[:-4] for f in os.listdir(os.path.join(os.path.dirname(__file__), "golden", "unicode")) if f.endswith(".gaf"))


@pytest.mark.parametrize("name", UNICODE)
def test_py_filter_unicode_digits(golden, name):
    """golden/unicode: decimal columns written with non-ASCII digits and blanks (int(), float(), str.rstrip() take them).  The
    Python oracle does what the reference did; so does the product's host-side decision for such lines (svjg/filter.py: host_line):
    it raises the reference's exception, or rewrites the line into an ASCII spelling that classifies the same."""
    from svjg import filter as flt
    q, u = f"{golden}/quirks", f"{golden}/unicode"
    man = json.load(open(f"{u}/manifest.json"))[name]
    edges = O.load_edges(f"{q}/q_svs_edges.json")
    alt = O.load_alt_node_len(f"{q}/q.gfa")
    lines = _read_lines(f"{u}/{name}.gaf")
    if man["rc"] == 0:
        D = O.classify(lines, edges, alt)
        assert O.dump_informative(D) == open(f"{u}/{name}.ref.json").read()
        ascii_lines = [flt.host_line(x).decode("utf-8") for x in lines]   # (ASCII but for a tag value inside the path column, which stays as it is)
        D2 = O.classify(ascii_lines, edges, alt)
        assert {k: [len(v[0]), len(v[1])] for k, v in D2.items()} == {k: [len(v[0]), len(v[1])] for k, v in D.items()}
    else:
        with pytest.raises(Exception) as ei:
            O.classify(lines, edges, alt)
        assert type(ei.value).__name__ == man["error"]
        with pytest.raises(Exception) as ei:                                # (from the host's own int() / float(), or from what the rewritten lines classify to)
            O.classify([flt.host_line(x).decode("utf-8") for x in lines], edges, alt)
        assert type(ei.value).__name__ == man["error"]


DOVER = sorted(f[:-4] for f in os.listdir(os.path.join(os.path.dirname(__file__), "golden", "dover")) if f.endswith(".gaf"))


@pytest.mark.parametrize("name", DOVER)
def test_py_filter_with_the_dover_flag(golden, name):
    """golden/dover: the reference run WITH -O 50 (d_over is then the list ["50"], filter-alignments.py:52-57, :88): TypeError at the
    first link with a candidate SV whose left sum can be formed (:153 -> :269), whatever comes first in the file otherwise, `{}` when
    there is no such link."""
    q, d = f"{golden}/quirks", f"{golden}/dover"
    man = json.load(open(f"{d}/manifest.json"))[name]
    edges = O.load_edges(f"{q}/q_svs_edges.json")
    alt = O.load_alt_node_len(f"{q}/q.gfa")
    lines = _read_lines(f"{d}/{name}.gaf")
    if man["rc"] == 0:
        assert O.dump_informative(O.classify(lines, edges, alt, d_over=["50"])) == open(f"{d}/{name}.ref.json").read() == "{}"
    else:
        with pytest.raises(Exception) as ei:
            O.classify(lines, edges, alt, d_over=["50"])
        assert type(ei.value).__name__ == man["error"]
        if man["error"] == "TypeError":
            from svjg import capi
            assert man["message"] == "TypeError: " + capi.DOVER_TYPE_ERROR and str(ei.value) == capi.DOVER_TYPE_ERROR


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


def test_py_graph_without_svs(golden):
    """golden/nosv: a VCF none of whose records becomes an SV of the graph (reference's constructor: `{}`), reference outputs `{}`
    and ./. for every row."""
    t = f"{golden}/nosv"
    man = json.load(open(f"{t}/manifest.json"))
    edges = O.load_edges(f"{t}/nosv_svs_edges.json")
    assert edges == {} == man["edges"]
    D = O.classify(_read_lines(f"{t}/nosv.gaf"), edges, O.load_alt_node_len(f"{t}/nosv.gfa"))
    assert O.dump_informative(D) == open(f"{t}/nosv.ref.json").read() == "{}"
    text, n = O.genotype_vcf(_read_lines(f"{t}/nosv.vcf"), D)
    assert text == open(f"{t}/nosv.ref_genotype.vcf").read() and f"Genotyped svs: {n}\n" == man["genotype_stdout"]


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


# ------------------------------------------------------------------------------------------------
# C oracle
# ------------------------------------------------------------------------------------------------
from oracle import oracle_c as OC  # noqa: E402


@pytest.mark.parametrize("name", QUIRKS)
def test_c_filter_quirks(golden, name):
    q = f"{golden}/quirks"
    man = json.load(open(f"{q}/manifest.json"))[name]
    orc = OC.COracle(O.load_edges(f"{q}/q_svs_edges.json"), O.load_alt_node_len(f"{q}/q.gfa"))
    raw = open(f"{q}/{name}.gaf", "rb").read()
    if man["rc"] == 0:
        counts, hits, n_lines = orc.filter(raw)
        D = OC.informative_dict(orc.sv_ids, hits, raw)
        assert O.dump_informative(D) == open(f"{q}/{name}.ref.json").read()
        ref = json.load(open(f"{q}/{name}.ref.json"))
        for i, sv in enumerate(orc.sv_ids):
            assert [int(counts[i, 0]), int(counts[i, 1])] == [len(x) for x in ref.get(sv, [[], []])]
    else:
        with pytest.raises(Exception) as ei:
            orc.filter(raw)
        assert type(ei.value).__name__ == man["error"]


def test_c_testdir(golden):
    t = f"{golden}/testdir"
    orc = OC.COracle(O.load_edges(f"{t}/test_svs_edges.json"), O.load_alt_node_len(f"{t}/test.gfa"))
    raw = open(f"{t}/test.gaf", "rb").read()
    counts, hits, n_lines = orc.filter(raw)
    assert n_lines == raw.count(b"\n")
    D = OC.informative_dict(orc.sv_ids, hits, raw)
    assert O.dump_informative(D) == open(f"{t}/ref_informative_aln.json").read()


@pytest.mark.parametrize("tag", ["g6_mixed", "g6_del"])
def test_c_synth_g6(golden, tag, tmp_path):
    """Medium synthetic case: inputs regenerated from the seed, reference outputs pinned by sha256 + counts."""
    import hashlib
    import synth
    g6 = json.load(open(f"{golden}/synth/g6.json"))[tag]
    pre = str(tmp_path / "s")
    synth.generate(prefix=pre, **g6["args"])
    for ext, sha in g6["sha256_inputs"].items():
        assert hashlib.sha256(open(pre + ext, "rb").read()).hexdigest() == sha, f"generator drifted: {ext}"
    orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
    raw = open(pre + ".gaf", "rb").read()
    counts, hits, n_lines = orc.filter(raw)
    got = {sv: [int(counts[i, 0]), int(counts[i, 1])] for i, sv in enumerate(orc.sv_ids) if counts[i].sum()}
    assert got == g6["counts"]
    js = O.dump_informative(OC.informative_dict(orc.sv_ids, hits, raw))
    assert hashlib.sha256(js.encode()).hexdigest() == g6["sha256_json"]
    text, n = O.genotype_vcf(open(pre + ".vcf").readlines(), json.loads(js))
    assert hashlib.sha256(text.encode()).hexdigest() == g6["sha256_vcf"]
    assert text == open(f"{golden}/synth/{tag}.ref_genotype.vcf").read()
    assert f"Genotyped svs: {n}\n" == g6["stdout"]


def test_realshape_lines(golden):
    """golden/realshape (lines shaped like real minigraph output, make_golden.py realshape): both oracles reproduce the
    reference's JSON / counts and its genotyped VCF."""
    import gzip
    from oracle import oracle_c as OC
    r = f"{golden}/realshape"
    edges = O.load_edges(f"{r}/r_svs_edges.json")
    alt = O.load_alt_node_len(f"{r}/r.gfa")
    ref_text = gzip.open(f"{r}/r.ref.json.gz", "rt").read()
    D = O.classify(_read_lines(f"{r}/r.gaf"), edges, alt)
    assert O.dump_informative(D) == ref_text
    text, n = O.genotype_vcf(_read_lines(f"{r}/r.vcf"), D, 1)
    assert text == open(f"{r}/r.ref_genotype.vcf").read() and n > 400
    orc = OC.COracle(edges, alt)
    counts, _, n_lines = orc.filter(open(f"{r}/r.gaf", "rb").read(), want_hits=False)
    ref = json.loads(ref_text)
    assert {sv: [int(counts[i, 0]), int(counts[i, 1])] for i, sv in enumerate(orc.sv_ids) if counts[i].sum()} == \
        {k: [len(v[0]), len(v[1])] for k, v in ref.items()}
    assert n_lines == len(_read_lines(f"{r}/r.gaf"))


def test_py_vcf_fuzz(golden):
    """golden/vcffuzz: 260 small VCFs of mutated rows through the reference's predict-genotype.py — the Python oracle writes the same
    text / stdout or dies with the same exception class."""
    cases = json.load(open(f"{golden}/vcffuzz/cases.json"))
    n_ok = 0
    for c in cases:
        D = {k: [["x\n"] * a, ["y\n"] * b] for k, (a, b) in c["counts"].items()}
        lines = c["vcf"].splitlines(keepends=True)
        if c["rc"] == 0:
            text, n = O.genotype_vcf(lines, D, min_support=c["minsupport"])
            assert text == c["out"] and f"Genotyped svs: {n}\n" == c["stdout"]
            n_ok += 1
        else:
            with pytest.raises(Exception) as ei:
                O.genotype_vcf(lines, D, min_support=c["minsupport"])
            assert type(ei.value).__name__ == c["error"]
    assert n_ok > 150


def _graphfuzz_case(c):
    import hashlib
    from tests import graph_fuzz
    edges, alt, lines = graph_fuzz.make_case(c["seed"], c["n_lines"])
    h = hashlib.sha256((json.dumps(edges, sort_keys=True) + json.dumps(alt, sort_keys=True) + "".join(lines)).encode()).hexdigest()
    if h != c["inputs_sha256"]:
        pytest.skip("this interpreter's random module does not reproduce the generator's stream")
    return edges, alt, lines


def test_graphfuzz_through_the_reference(golden):
    """golden/graphfuzz: 40 random graphs (hazard-prone names, multi-SV links, links in both reading directions, hubs) with 500 random
    walks each, through the reference's filter-alignments.py: both oracles count what it counted, the Python oracle's JSON text has its
    sha256."""
    import hashlib
    from oracle import oracle_c as OC
    for c in json.load(open(f"{golden}/graphfuzz/cases.json")):
        edges, alt, lines = _graphfuzz_case(c)
        D = O.classify(lines, edges, alt)
        assert {k: list(v) for k, v in O.counts_of(D).items()} == c["counts"], c["seed"]
        assert hashlib.sha256(O.dump_informative(D).encode()).hexdigest() == c["json_sha256"], c["seed"]
        orc = OC.COracle(edges, alt)
        want, _, n = orc.filter("".join(lines).encode(), want_hits=False)
        assert {sv: [int(want[i, 0]), int(want[i, 1])] for i, sv in enumerate(orc.sv_ids) if want[i].sum()} == c["counts"] and n == c["n_lines"]


def _longpath_case(c):
    import hashlib
    from tests import longpath_fuzz
    edges, alt, lines, fatal = longpath_fuzz.make_case(c["seed"], c["n_long"], c["n_fatal"])
    h = hashlib.sha256((json.dumps(edges, sort_keys=True) + json.dumps(alt, sort_keys=True) + "".join(lines) + "".join(fatal)).encode()).hexdigest()
    if h != c["inputs_sha256"]:
        pytest.skip("this interpreter's random module does not reproduce the generator's stream")
    return edges, alt, lines, fatal


def test_longpath_through_the_reference(golden):
    """golden/longpath (r05): 40 walks of 65..216 nodes over graphs of >= 2 000 nodes — clean for 64 nodes, then one late event (a name
    the graph lacks, of positive / negative / 5 Gbp length; a name of 49+ bytes; a hazard name; a 40 Mbp node; a revisit; ids that turn;
    another contig; a stretch walked back) — through the reference's filter-alignments.py: both oracles count what it counted, the
    Python oracle's JSON text has its sha256, and a line with an insertion node the GFA lacks kills both with the reference's KeyError."""
    import hashlib
    from oracle import oracle_c as OC
    for c in json.load(open(f"{golden}/longpath/cases.json")):
        edges, alt, lines, fatal = _longpath_case(c)
        D = O.classify(lines, edges, alt)
        assert {k: list(v) for k, v in O.counts_of(D).items()} == c["counts"], c["seed"]
        assert hashlib.sha256(O.dump_informative(D).encode()).hexdigest() == c["json_sha256"], c["seed"]
        orc = OC.COracle(edges, alt)
        want, _, n = orc.filter("".join(lines).encode(), want_hits=False)
        assert {sv: [int(want[i, 0]), int(want[i, 1])] for i, sv in enumerate(orc.sv_ids) if want[i].sum()} == c["counts"] and n == c["n_lines"]
        for f, err in zip(fatal, c["fatal_errors"]):
            bad = lines[:4] + [f] + lines[4:6]
            for run in (lambda: O.classify(bad, edges, alt), lambda: orc.filter("".join(bad).encode(), want_hits=False)):
                with pytest.raises(Exception) as ei:
                    run()
                assert type(ei.value).__name__ == err


@pytest.mark.parametrize("tag", ["hla", "ucsc"])
def test_contig_names_of_the_grch38_analysis_set(golden, tag):
    """golden/contigs (r05): HLA-DRB1*15:03:01:01, HLA-A*01:01:01:01 (':' '*' '-' inside the contig part), chrUn_JTFH01001998v1_decoy,
    chr6_GL000250v2_alt, chrEBV — the reference's JSON for 90 walks per graph; both oracles, both graph loaders and the exact routine
    (host build) reproduce it."""
    from oracle import oracle_c as OC
    from svjg.graph import Graph
    from tests.hostsim import sim
    pre = f"{golden}/contigs/{tag}"
    ref_text = open(pre + ".ref.json").read()
    ref = {k: [len(v[0]), len(v[1])] for k, v in json.loads(ref_text).items()}
    edges, alt = O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa")
    lines = open(pre + ".gaf").read().splitlines(True)
    D = O.classify(lines, edges, alt)
    assert {k: list(v) for k, v in O.counts_of(D).items()} == ref and O.dump_informative(D) == ref_text
    orc = OC.COracle(edges, alt)
    want, _, n = orc.filter(open(pre + ".gaf", "rb").read(), want_hits=False)
    assert {sv: [int(want[i, 0]), int(want[i, 1])] for i, sv in enumerate(orc.sv_ids) if want[i].sum()} == ref and n == len(lines)
    for native in (False, True):
        g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa", native=native)
        assert sim.check_tables(g) == 0
        assert g.n_hazard == 0                                    # (r05: a ':' inside a contig name no longer marks every node name hazard-prone)
        for wave in (0, 2, 5):
            counts, n_lines = sim.classify(g, open(pre + ".gaf", "rb").read(), True, wave)
            assert n_lines == n and {g.sv_ids[i]: [int(counts[i, 0]), int(counts[i, 1])] for i in range(g.n_slots) if counts[i].sum()} == ref


def _longtail_case(c, tmp_path):
    import hashlib
    import synth
    from tests import longpath_fuzz
    pre = str(tmp_path / f"s{c['seed']}")
    n_aln, n_sv, n_chrom, mix, seed = c["synth"]
    inf = synth.generate(pre, n_aln, n_sv, n_chrom, mix, seed, write_gaf=False, return_gaf=True)
    base = inf["gaf"].tobytes().split(b"\n")[:-1]
    text, fatal = longpath_fuzz.make_tail_case(c["tail_seed"], base, c["n_mut"])
    fatal = fatal[:5]
    if hashlib.sha256(text + b"".join(fatal)).hexdigest() != c["inputs_sha256"]:
        pytest.skip("this interpreter's random module does not reproduce the generator's stream")
    return pre, text, fatal


def test_longtail_through_the_reference(golden, tmp_path):
    """golden/longtail (r05): lines longer than 8 KB with one event in a 6..40 KB tail at a boundary position (a carriage return — the
    reference's universal newlines end the line there —, "d:", an id:f: tag, bytes >= 0x80, the terminator, none) through the reference's
    filter-alignments.py: both oracles and the exact routine (host build) count what it counted, the Python oracle's JSON has its sha256,
    an id:f: tag with a malformed value in a tail kills all three with the reference's ValueError."""
    import hashlib
    from oracle import oracle_c as OC
    from svjg.graph import Graph
    from tests.hostsim import sim
    for c in json.load(open(f"{golden}/longtail/cases.json")):
        pre, text, fatal = _longtail_case(c, tmp_path)
        edges, alt = O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa")
        D = O.classify(text.decode().splitlines(True), edges, alt)
        assert {k: list(v) for k, v in O.counts_of(D).items()} == c["counts"], c["seed"]
        assert hashlib.sha256(O.dump_informative(D).encode()).hexdigest() == c["json_sha256"], c["seed"]
        orc = OC.COracle(edges, alt)
        want, _, n = orc.filter(text, want_hits=False)
        assert {sv: [int(want[i, 0]), int(want[i, 1])] for i, sv in enumerate(orc.sv_ids) if want[i].sum()} == c["counts"]
        g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
        counts, n2 = sim.classify(g, text, True, 5)
        assert n2 == n and {g.sv_ids[i]: [int(counts[i, 0]), int(counts[i, 1])] for i in range(g.n_slots) if counts[i].sum()} == c["counts"]
        for f, err in zip(fatal, c["fatal_errors"]):
            for run in (lambda: O.classify(f.decode().splitlines(True), edges, alt), lambda: orc.filter(f, want_hits=False), lambda: sim.classify(g, f, True, 0)):
                with pytest.raises(Exception) as ei:
                    run()
                assert type(ei.value).__name__ == err
