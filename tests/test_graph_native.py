"""The native graph loader (libsvjg_host.so: svjg_graph_load) is the fast path of svjg/graph.py: on every input it
accepts it must build exactly the tables the Python loader builds, and it must leave everything else to Python."""
import json
import os

import numpy as np
import pytest

from svjg.graph import Graph

FIELDS = ("nodes", "edges", "hits", "chrom_off", "chrom_lo", "chrom_hash")
SCALARS = ("n_nodes", "n_edges", "n_hits", "n_slots", "n_hazard", "chrom_names", "chroms", "sv_ids", "slot_of", "d_over", "flags")


def _same(a, b):
    for f in FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), f
    for f in SCALARS:
        assert getattr(a, f) == getattr(b, f), f
    assert a.node_names == b.node_names


@pytest.mark.parametrize("case", ["quirks", "testdir"])
def test_golden_graphs(golden, case):
    ej, gfa = {"quirks": ("quirks/q_svs_edges.json", "quirks/q.gfa"), "testdir": ("testdir/test_svs_edges.json", "testdir/test.gfa")}[case]
    py = Graph.from_files(f"{golden}/{ej}", f"{golden}/{gfa}", native=False)
    nat = Graph.from_files(f"{golden}/{ej}", f"{golden}/{gfa}", native=True)
    _same(py, nat)
    assert nat.n_hazard == py.n_hazard


@pytest.mark.parametrize("mix,n_sv,n_chrom,seed", [("mixed", 3000, 3, 5), ("del", 500, 1, 9), ("mixed", 1200, 12, 11)])
def test_synthetic_graphs(tmp_path, mix, n_sv, n_chrom, seed):
    import synth
    pre = str(tmp_path / "g")
    synth.generate(pre, 0, n_sv, n_chrom, mix, seed, write_gaf=False)
    _same(Graph.from_files(pre + "_svs_edges.json", pre + ".gfa", native=False, all_slow=True),
          Graph.from_files(pre + "_svs_edges.json", pre + ".gfa", native=True, all_slow=True))


def test_layouts_and_empty(tmp_path):
    """compact JSON, an empty value list, an empty table; a graph with hazard names ('1' / '11', '.1' / '.10')."""
    d = {"1:1-100@+@1:101-200@+": [["1:DEL-100-150", 0], ["1:INS-100-1", 0], ["1:INS-100-2", 0]],
         "11:1-100@+@11:101-200@-": [["11:INV-100-200", 1]], "1:101-200@-@1:1-100@-": [["1:X", 1]],
         "1:100.1@+@1:101-200@+": [["1:INS-100-1", 1]], "1:100.10@+@1:101-200@+": [["1:INS-100-10", 1]], "2:5-9@+@2:10-20@+": []}
    gfa = "H\tVN:Z:1.0\nS\t1:100.1\tACGTACGT\nS\t1:100.10\tACG\textra\nS\t1:1-100\t*\nL\t1:1-100\t+\t1:101-200\t+\t0M\n"
    open(tmp_path / "g.gfa", "w").write(gfa)
    for name, text in (("a", json.dumps(d)), ("b", json.dumps(d, sort_keys=True, indent=4)), ("c", "{}"), ("d", " {\n} \n")):
        p = str(tmp_path / f"{name}_svs_edges.json")
        open(p, "w").write(text)
        _same(Graph.from_files(p, str(tmp_path / "g.gfa"), native=False), Graph.from_files(p, str(tmp_path / "g.gfa"), native=True))
    g = Graph.from_files(str(tmp_path / "a_svs_edges.json"), str(tmp_path / "g.gfa"), native=True)
    assert g.n_hazard > 0 and g.n_hits == 8          # two merged queries of four hits each


def test_irregular_inputs_are_left_to_python(tmp_path):
    from svjg import capi
    gfa = str(tmp_path / "g.gfa")
    open(gfa, "w").write("S\t1:5.1\tACGT\n")
    ok = '{"1:1-4@+@1:5-9@+": [["1:DEL-4-9", 0]]}'
    cases = {
        "escape": '{"1:1-4@+@1:5-9@+": [["1:DEL\\u002d4-9", 0]]}',
        "unicode": '{"1:1-4@+@1:5-9@+": [["é:DEL-4-9", 0]]}',
        "duplicate": '{"1:1-4@+@1:5-9@+": [["a", 0]], "1:1-4@+@1:5-9@+": [["b", 0]]}',
        "duplicate_then_empty": '{"1:1-4@+@1:5-9@+": [["a", 0]], "1:1-4@+@1:5-9@+": []}',
        "empty_then_duplicate": '{"1:1-4@+@1:5-9@+": [], "1:1-4@+@1:5-9@+": [["a", 0]]}',
        "allele2": '{"1:1-4@+@1:5-9@+": [["a", 2]]}',
        "allele_true": '{"1:1-4@+@1:5-9@+": [["a", true]]}',
        "allele_float": '{"1:1-4@+@1:5-9@+": [["a", 0.0]]}',
        "leading_zero": '{"1:01-4@+@1:5-9@+": [["a", 0]]}',
        "no_colon": '{"x@+@1:5-9@+": [["a", 0]]}',
        "bad_strand": '{"1:1-4@*@1:5-9@+": [["a", 0]]}',
        "five_parts": '{"1:1-4@+@1:5-9@+@x": [["a", 0]]}',
        "end_before_start": '{"1:9-4@+@1:5-9@+": [["a", 0]]}',
        "same_coordinate": '{"1:1-4@+@1:1-9@+": [["a", 0]]}',
        "trailing": ok + " x",
        "not_a_dict": "[]",
    }
    for name, text in cases.items():
        p = str(tmp_path / f"{name}.json")
        open(p, "w", encoding="utf-8").write(text)
        assert capi.graph_load_native(p, gfa) is None, name
    p = str(tmp_path / "ok.json")
    open(p, "w").write(ok)
    assert capi.graph_load_native(p, gfa) is not None
    for name, g in {"cr": "S\t1:5.1\tACGT\r\n", "two_columns": "S\t1:5.1\n", "empty_sequence": "S\t1:5.1\t\n", "utf8": "S\t1:5.1\tACGé\n"}.items():
        q = str(tmp_path / f"{name}.gfa")
        open(q, "w", encoding="utf-8", newline="").write(g)
        assert capi.graph_load_native(p, q) is None, name
    with pytest.raises(OSError):
        capi.graph_load_native(str(tmp_path / "missing.json"), gfa)


@pytest.mark.parametrize("seed", range(10))
def test_random_graphs(tmp_path, seed):
    """tests/graph_fuzz.py's graphs (hazard-prone names, multi-SV links, links in both directions, hubs) written as the files the
    reference's constructor writes (indent-4 JSON, GFA with the insertion nodes' sequences): native and Python loader build the same
    tables, and those are the tables of the in-memory constructor."""
    from tests import graph_fuzz
    edges, alt, _ = graph_fuzz.make_case(300 + seed, 1)
    p = str(tmp_path / "g_svs_edges.json")
    open(p, "w").write(json.dumps(edges, indent=4))
    with open(tmp_path / "g.gfa", "w") as fh:
        fh.write("H\tVN:Z:1.0\n")
        names = sorted({n for k in edges for n in (k.split("@")[0], k.split("@")[2])} | set(alt))
        for n in names:
            fh.write(f"S\t{n}\t{'ACGT' * (alt[n] // 4) + 'A' * (alt[n] % 4) if n in alt else '*'}\n")
    py = Graph.from_files(p, str(tmp_path / "g.gfa"), native=False)
    nat = Graph.from_files(p, str(tmp_path / "g.gfa"), native=True)
    _same(py, nat)
    _same(py, Graph(edges, alt))
    assert py.n_hazard > 0 or seed >= 0


def _brute_hazards(names):
    return {x for x in names for y in names if x != y and x in y}


def test_hazard_names_with_colons_in_contig_names(tmp_path):
    """r05: which node names are proper substrings of other node names (the strand quirk of filter-alignments.py:206 — such names take the
    exact routine) is now decided exactly for ANY contig names: a ':' inside a contig name (HLA-DRB1*15:03:01:01) used to mark every node
    of the graph.  Both loaders against the brute-force definition on graphs whose contig names nest in every way the rule has to see —
    a contig that is the tail of another's name, a node name that sits at an INNER colon of a longer contig name — and on golden/contigs."""
    import json
    import random
    from svjg.graph import Graph
    rng = random.Random(5)
    chroms = ["7", "y:7", "HLA-A*01:01:01:01", "01", "p", "p:3-4q", "q:12", "12", "chr6", "chr6:1-2x:chr6", "1", "11", "x:1"]
    edges, alt = {}, {}
    for c in chroms:
        cuts = sorted(rng.sample(range(2, 60), 4))
        if c in ("p", "7", "12", "1", "chr6"):
            cuts = [4, 9, 20, 33]                                   # (the same breakpoints on the contigs that nest: "7:5-9" inside "y:7:5-9")
        if c in ("y:7", "q:12", "11", "x:1", "chr6:1-2x:chr6"):
            cuts = [4, 9, 20, 33]
        starts, ends = [1] + [x + 1 for x in cuts], cuts + [cuts[-1] + 50]
        nodes = [f"{c}:{a}-{b}" for a, b in zip(starts, ends)]
        if c == "p":
            nodes[1] = "p:3-4"                                       # "p:3-4" stands inside "p:3-4q:..." at an inner colon
            nodes[0] = "p:1-2"
        for i in range(len(nodes) - 1):
            edges[f"{nodes[i]}@+@{nodes[i + 1]}@+"] = [[f"{c}:DEL-{i}-{i + 60}", 0]]
        an = f"{c}:{starts[2]}.1"
        alt[an] = 77
        edges[f"{nodes[1]}@+@{an}@+"] = [[f"{c}:INS-{starts[2]}-1", 1]]
        edges[f"{an}@+@{nodes[2]}@+"] = [[f"{c}:INS-{starts[2]}-1", 1]]
    names = {x for k in edges for x in (k.split("@")[0], k.split("@")[2])} | set(alt)
    want = _brute_hazards(names)
    assert {"7:5-9", "p:3-4", "12:5-9", "1:5-9", "chr6:5-9", "1:10.1"} <= want and len(want) < len(names) / 2
    with open(tmp_path / "g_svs_edges.json", "w") as fh:
        fh.write(json.dumps(edges, indent=4))
    with open(tmp_path / "g.gfa", "w") as fh:
        fh.write("H\tVN:Z:1.0\n" + "".join(f"S\t{n}\t{'A' * alt[n] if n in alt else '*'}\n" for n in sorted(names)))
    for native in (False, True):
        g = Graph.from_files(str(tmp_path / "g_svs_edges.json"), str(tmp_path / "g.gfa"), native=native)
        got = {g.node_names[i] for i in range(g.n_nodes) if int(g.nodes["row"][i]) & 0x80000000}
        assert got == want, (native, sorted(got ^ want))
    # the fixture with GRCh38 analysis-set names: no node name of it stands inside another
    gold = os.path.join(os.path.dirname(__file__), "golden", "contigs")
    for native in (False, True):
        g = Graph.from_files(f"{gold}/hla_svs_edges.json", f"{gold}/hla.gfa", native=native)
        assert g.n_hazard == len(_brute_hazards(set(g.node_names))) == 0
    # ... and without a colon in a contig name the general rule is the grouped one (random hazard-prone graphs)
    from tests import graph_fuzz
    for seed in range(6):
        e2, a2, _ = graph_fuzz.make_case(seed, 1)
        g = Graph(e2, a2)
        info = {n: None for n in g.node_names}
        got = {g.node_names[i] for i in range(g.n_nodes) if int(g.nodes["row"][i]) & 0x80000000}
        assert got == _brute_hazards(set(g.node_names)) == Graph._hazards_general(info, g.chroms)
