"""CPU-side check of the per-line device routines (svjedi-graph_amd/csrc/svjg_line.h, compiled with g++ by
tests/hostsim) and of the host graph tables against the golden vectors and the C oracle.
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


@pytest.mark.parametrize("force_slow", [False, True])
@pytest.mark.parametrize("name", QUIRKS)
def test_quirks(golden, name, force_slow):
    q = f"{golden}/quirks"
    man = json.load(open(f"{q}/manifest.json"))[name]
    g = Graph.from_files(f"{q}/q_svs_edges.json", f"{q}/q.gfa")
    raw = open(f"{q}/{name}.gaf", "rb").read()
    if man["rc"] == 0:
        counts, n_lines, n_def = sim.classify(g, raw, force_slow)
        ref = json.load(open(f"{q}/{name}.ref.json"))
        assert _as_dict(g, counts) == {k: [len(v[0]), len(v[1])] for k, v in ref.items()}
    else:
        with pytest.raises(Exception) as ei:
            sim.classify(g, raw, force_slow)
        assert type(ei.value).__name__ == man["error"]


@pytest.mark.parametrize("force_slow", [False, True])
def test_testdir(golden, force_slow):
    t = f"{golden}/testdir"
    g = Graph.from_files(f"{t}/test_svs_edges.json", f"{t}/test.gfa")
    raw = open(f"{t}/test.gaf", "rb").read()
    counts, n_lines, n_def = sim.classify(g, raw, force_slow)
    ref = json.load(open(f"{t}/ref_informative_aln.json"))
    assert _as_dict(g, counts) == {k: [len(v[0]), len(v[1])] for k, v in ref.items()}
    if not force_slow:
        assert n_def == 0          # a regular graph never needs the exact path


@pytest.mark.parametrize("tag", ["g6_mixed", "g6_del"])
def test_synth(golden, tag, tmp_path):
    import synth
    g6 = json.load(open(f"{golden}/synth/g6.json"))[tag]
    pre = str(tmp_path / "s")
    synth.generate(prefix=pre, **g6["args"])
    g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    raw = open(pre + ".gaf", "rb").read()
    counts, n_lines, n_def = sim.classify(g, raw)
    assert _as_dict(g, counts) == g6["counts"]
    assert n_lines == g6["args"]["n_aln"]
    assert 0 < n_def < n_lines // 100      # only the lines with a revisited node take the exact path
    orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
    cut = raw[: 2_000_000]
    cut = cut[: cut.rfind(b"\n") + 1]
    c_or, _, _ = orc.filter(cut, want_hits=False)
    counts3, _, _ = sim.classify(g, cut, force_slow=True)
    assert _as_dict(g, counts3) == {sv: [int(c_or[i, 0]), int(c_or[i, 1])] for i, sv in enumerate(orc.sv_ids) if c_or[i].sum()}
