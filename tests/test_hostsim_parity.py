"""CPU-side check of the exact per-line device routine and the table lookups (svjedi-graph_amd/csrc/svjg_line.h,
compiled with g++ by tests/hostsim) and of the host graph tables, against the golden vectors and the C oracle.
The kernels themselves are checked by tests/test_gpu_parity.py (-m gpu)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_c as OC
from oracle import oracle_py as O
from svjg.graph import Graph
from tests.hostsim import sim

QUIRKS = sorted(f[:-4] for f in os.listdir(os.path.join(os.path.dirname(__file__), "golden", "quirks"))
                if f.endswith(".gaf"))


def _as_dict(graph, counts):
    return {graph.sv_ids[i]: [int(counts[i, 0]), int(counts[i, 1])] for i in range(graph.n_slots) if counts[i].sum()}


@pytest.mark.parametrize("tables,wave", [(True, 0), (False, 0), (True, 1), (True, 2), (True, 3), (True, 4), (True, 5)], ids=["name_table", "sorted_table", "wave_lanes", "wave_two_phase", "lane_cached", "wave_two_phase_every_offset", "wave_strands_by_id"])
@pytest.mark.parametrize("name", QUIRKS)
def test_quirks(golden, name, tables, wave):
    q = f"{golden}/quirks"
    man = json.load(open(f"{q}/manifest.json"))[name]
    g = Graph.from_files(f"{q}/q_svs_edges.json", f"{q}/q.gfa")
    raw = open(f"{q}/{name}.gaf", "rb").read()
    if man["rc"] == 0:
        counts, n_lines = sim.classify(g, raw, tables, wave)
        ref = json.load(open(f"{q}/{name}.ref.json"))
        assert _as_dict(g, counts) == {k: [len(v[0]), len(v[1])] for k, v in ref.items()}
    else:
        with pytest.raises(Exception) as ei:
            sim.classify(g, raw, tables, wave)
        assert type(ei.value).__name__ == man["error"]


DOVER = sorted(f[:-4] for f in os.listdir(os.path.join(os.path.dirname(__file__), "golden", "dover")) if f.endswith(".gaf"))


@pytest.mark.parametrize("wave", [0, 1, 2, 3], ids=["one_lane", "wave_lanes", "wave_two_phase", "lane_cached"])
@pytest.mark.parametrize("name", DOVER)
def test_dover_flag(golden, name, wave):
    """golden/dover (the reference run with -O): the exact routine under SVJG_GRAPH_DOVER_LIST raises TypeError where the reference
    does — after the left sum's node lengths, before the right sum — in all three of its forms"""
    from svjg.graph import GRAPH_DOVER_LIST
    q, d = f"{golden}/quirks", f"{golden}/dover"
    man = json.load(open(f"{d}/manifest.json"))[name]
    g = Graph.from_files(f"{q}/q_svs_edges.json", f"{q}/q.gfa")
    g.flags |= GRAPH_DOVER_LIST
    raw = open(f"{d}/{name}.gaf", "rb").read()
    if man["rc"] == 0:
        counts, n_lines = sim.classify(g, raw, True, wave)
        assert counts.sum() == 0 and n_lines == man["n_lines"]
    else:
        with pytest.raises(Exception) as ei:
            sim.classify(g, raw, True, wave)
        assert type(ei.value).__name__ == man["error"]


@pytest.mark.parametrize("tables,wave", [(True, 0), (True, 2), (True, 3), (True, 5)], ids=["name_table", "wave_two_phase", "lane_cached", "wave_strands_by_id"])
def test_realshape_lines(golden, tables, wave):
    """the exact per-line routine on the lines shaped like real minigraph output (paths of up to 300 nodes, kilobyte tags)"""
    import gzip
    r = f"{golden}/realshape"
    g = Graph.from_files(f"{r}/r_svs_edges.json", f"{r}/r.gfa")
    counts, n_lines = sim.classify(g, open(f"{r}/r.gaf", "rb").read(), tables, wave)
    ref = json.loads(gzip.open(f"{r}/r.ref.json.gz", "rt").read())
    assert _as_dict(g, counts) == {k: [len(v[0]), len(v[1])] for k, v in ref.items()}


def test_testdir(golden):
    t = f"{golden}/testdir"
    g = Graph.from_files(f"{t}/test_svs_edges.json", f"{t}/test.gfa")
    raw = open(f"{t}/test.gaf", "rb").read()
    counts, n_lines = sim.classify(g, raw)
    ref = json.load(open(f"{t}/ref_informative_aln.json"))
    assert _as_dict(g, counts) == {k: [len(v[0]), len(v[1])] for k, v in ref.items()}
    assert sim.check_tables(g) == 0


@pytest.mark.parametrize("tag", ["g6_mixed", "g6_del"])
def test_synth(golden, tag, tmp_path):
    import synth
    g6 = json.load(open(f"{golden}/synth/g6.json"))[tag]
    pre = str(tmp_path / "s")
    synth.generate(prefix=pre, **g6["args"])
    g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    assert sim.check_tables(g) == 0          # the kernel's name / link hash tables agree with the node table and CSR rows
    raw = open(pre + ".gaf", "rb").read()
    cut = raw[: 3_000_000]
    cut = cut[: cut.rfind(b"\n") + 1]
    counts, n_lines = sim.classify(g, cut)
    orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
    c_or, _, n2 = orc.filter(cut, want_hits=False)
    assert n_lines == n2
    assert _as_dict(g, counts) == {sv: [int(c_or[i, 0]), int(c_or[i, 1])] for i, sv in enumerate(orc.sv_ids) if c_or[i].sum()}


def test_byte_classes_by_bit_planes():
    """svjg_planes.h (phase B1 of the main kernel): the masks the transposed planes give are those of a byte loop —
    every byte value at every position of a span, GAF-like text, random bytes."""
    rng = np.random.default_rng(7)
    every = np.concatenate([np.roll(np.arange(256, dtype=np.uint8), k) for k in range(64)])          # 256 spans: every value at every position
    gaf = np.frombuffer((b"read/1\t1000\t10\t990\t+\t>chr1:0-5000<chr1:5000.1>chr1:5000-9000\t9000\t1\t981\t900\t980\t60\tid:f:0.9\tcg:Z:5=\r\n" * 40)[:64 * 60], dtype=np.uint8)
    rnd = rng.integers(0, 256, 64 * 4096, dtype=np.uint8)
    few = rng.choice(np.frombuffer(b"\n\r\t<>=;:d09/e", dtype=np.uint8), 64 * 1024)
    text = np.concatenate([every, gaf, rnd, few])
    got = sim.span_classes(text.tobytes())
    b = text.reshape(-1, 64)
    wt = np.uint64(1) << np.arange(64, dtype=np.uint64)

    def mask(sel):
        return (sel.astype(np.uint64) * wt).sum(axis=1, dtype=np.uint64)
    digit = (b >= 0x30) & (b <= 0x39)
    want = [b == 0x0A, b == 0x0D, b == 0x09, (b == 0x3C) | (b == 0x3E), ~(digit | (b == 0x09)), b == 0x64, b == 0x3A, b >= 0x80]
    for k, sel in enumerate(want):
        assert np.array_equal(got[:, k], mask(sel)), k


@pytest.mark.parametrize("seed", range(8))
def test_random_graphs(seed):
    """tests/graph_fuzz.py: random graphs with hazard-prone names, multi-SV links, links in both reading directions, hub nodes; the
    kernels' tables agree with the node table and the CSR rows, and the exact routine (through the tables and without them) counts
    what the C oracle counts — which is what the Python oracle counts."""
    from tests import graph_fuzz
    edges, alt, lines = graph_fuzz.make_case(seed, 600)
    g = Graph(edges, alt)
    assert sim.check_tables(g) == 0
    text = "".join(lines).encode()
    orc = OC.COracle(edges, alt)
    want, _, n = orc.filter(text, want_hits=False)
    wd = {sv: [int(want[i, 0]), int(want[i, 1])] for i, sv in enumerate(orc.sv_ids) if want[i].sum()}
    assert wd == {k: list(v) for k, v in O.counts_of(O.classify(lines, edges, alt)).items()} and sum(map(sum, wd.values())) > 1000
    for tables in (True, False):
        counts, n_lines = sim.classify(g, text, tables=tables)
        assert n_lines == n and _as_dict(g, counts) == wd
    # ... and so do the forms the kernels run it in: 64 cooperating lanes, the two-phase wave routine with its tables of path pieces (one
    # candidate position per piece, asked of the eight bytes around the colons first) and without them, one lane with its per-node results kept
    for wave in (1, 2, 3, 4, 5):
        counts, n_lines = sim.classify(g, text, True, wave)
        assert n_lines == n and _as_dict(g, counts) == wd, wave


@pytest.mark.parametrize("seed", range(4))
def test_long_path_fuzz(seed):
    """tests/longpath_fuzz.py: graphs of >= 2 000 nodes, walks of 65..216 nodes that stay clean for 64 nodes and meet ONE late event (a
    name the graph lacks — of positive, negative, 5 Gbp length —, a name of 49+ bytes, a hazard name, a 40 Mbp node, a revisit, ids that
    turn, another contig, a stretch walked back): the two oracles agree, the exact routine in every form the kernels run it in counts the
    same, and dies with the same exception on a line with an insertion node the GFA lacks."""
    from tests import longpath_fuzz
    edges, alt, lines, fatal = longpath_fuzz.make_case(seed, 30, 3)
    g = Graph(edges, alt)
    assert sim.check_tables(g) == 0 and g.n_nodes >= 2000
    text = "".join(lines).encode()
    orc = OC.COracle(edges, alt)
    want, _, n = orc.filter(text, want_hits=False)
    wd = {sv: [int(want[i, 0]), int(want[i, 1])] for i, sv in enumerate(orc.sv_ids) if want[i].sum()}
    assert wd == {k: list(v) for k, v in O.counts_of(O.classify(lines, edges, alt)).items()} and sum(map(sum, wd.values())) > 2000
    for wave in (0, 2, 3, 5):
        counts, n_lines = sim.classify(g, text, True, wave)
        assert n_lines == n and _as_dict(g, counts) == wd, wave
    for f in fatal:
        bad = "".join(lines[:4] + [f] + lines[4:8])
        with pytest.raises(KeyError):
            O.classify(bad.splitlines(True), edges, alt)
        with pytest.raises(KeyError):
            orc.filter(bad.encode(), want_hits=False)
        for wave in (0, 2, 5):
            with pytest.raises(KeyError):
                sim.classify(g, bad.encode(), True, wave)


def test_tables_with_node_names_of_49_to_64_bytes(tmp_path):
    """r06: names of 49..64 bytes are in the main kernel's table (windows of the last 48 bytes + the first bytes in name_pfx, both hashed);
    beyond 64 bytes they stay out.  The table check finds every such node under its hash with its prefix words; the exact routine agrees
    with the C oracle on a graph whose contigs differ only in their first bytes."""
    import synth
    pre = str(tmp_path / "n")
    synth.generate(pre, 3000, 600, 7, "mixed", 57)
    a = "scaffold_of_an_assembly_that_names_them_at_length"
    ren = {"chr2": "chromosome_2", "chr3": "chr1_KI270706v1_random", "chr4": "a_contig_name_of_thirty_six_bytes_xx", "chr5": "A_" + a[:42], "chr6": "B_" + a[:42],
           "chr7": "a_contig_name_that_is_really_fifty_five_bytes_long_abcde"}
    for ext in (".gfa", "_svs_edges.json", ".gaf"):
        t = open(pre + ext).read()
        for old, new in ren.items():
            t = t.replace(old + ":", new + ":").replace(old + "\t", new + "\t")
        open(pre + ext, "w").write(t)
    g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    lens = np.array([len(n) for n in g.node_names])
    assert ((lens > 48) & (lens <= 64)).sum() > 100 and (lens > 64).sum() > 50
    assert sim.check_tables(g) == 0
    stats = sim.table_stats(g)
    assert stats[0] == 0 and stats[1] == int((lens > 64).sum())               # left out: none; skipped: exactly the names beyond 64 bytes
    orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
    gaf = np.fromfile(pre + ".gaf", dtype=np.uint8)
    want, _, _ = orc.filter(gaf, want_hits=False)
    order = [orc.sv_ids.index(x) for x in g.sv_ids]
    for wave in (0, 2, 5):
        c, _ = sim.classify(g, gaf, True, wave)
        assert (c.astype(np.uint64) == want[order]).all()
