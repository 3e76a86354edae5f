"""500 mutated GAF fragments, each evaluated by the reference itself (tests/golden/make_fuzz.py -> golden/fuzz/fuzz.json):
what the reference did with a fragment — the per-SV list lengths of its JSON, or the class of the exception it died with —
is what the CPU oracles and the exact per-line device routine (compiled for the host by tests/hostsim) must do too.
The HIP kernels take the same cases in tests/test_gpu_parity.py::test_fuzz_cases."""
import base64
import io
import json

import numpy as np
import pytest

from oracle import oracle_c as OC
from oracle import oracle_py as O
from svjg.graph import Graph
from tests.hostsim import sim


@pytest.fixture(scope="module")
def fuzz(golden):
    t = f"{golden}/testdir"
    cases = json.load(open(f"{golden}/fuzz/fuzz.json"))["cases"]
    for c in cases:
        c["raw"] = base64.b64decode(c["gaf"])
    edges, alt = O.load_edges(f"{t}/test_svs_edges.json"), O.load_alt_node_len(f"{t}/test.gfa")
    return cases, edges, alt, Graph.from_files(f"{t}/test_svs_edges.json", f"{t}/test.gfa")


def _want(c):
    return ("ok", c["counts"]) if c["rc"] == 0 else ("died", c["error"])


def documented_divergence(c):
    """decimal columns written with non-ASCII Unicode digits (int() takes them): the C oracle (test infrastructure) reads ASCII
    only; the device routine sets such a line aside for the host (tests/hostsim: HostLine), whose part is
    svjg/filter.py: resolve_host_lines (tests/test_oracle_golden.py::test_py_filter_unicode_digits, tests/test_gpu_parity.py)"""
    return b"\xd9\xa3" in c["raw"]


def test_python_oracle(fuzz):
    cases, edges, alt, _ = fuzz
    for i, c in enumerate(cases):
        try:
            lines = io.TextIOWrapper(io.BytesIO(c["raw"]), encoding="utf-8").readlines()      # text mode, universal newlines
            D = O.classify(lines, edges, alt)
            got = ("ok", {k: [len(v[0]), len(v[1])] for k, v in D.items()})
        except Exception as e:
            got = ("died", type(e).__name__)
        assert got == _want(c), (i, c["raw"])


def test_c_oracle_and_exact_device_routine(fuzz):
    cases, edges, alt, g = fuzz
    orc = OC.COracle(edges, alt)
    for i, c in enumerate(cases):
        if c["rc"] and c["error"] == "UnicodeDecodeError":
            continue                                            # raised by the text-mode read, in front of the per-line code
        if documented_divergence(c):
            # the device routine neither counts nor condemns the line: it asks the host (or meets another line's error first)
            for tables, wave in ((True, 0), (True, 1), (True, 2), (True, 3)):
                try:
                    sim.classify(g, c["raw"], tables, wave)
                    got = "ok"
                except sim.HostLine:
                    got = "host"
                except Exception as e:
                    got = type(e).__name__
                assert got == "host" or (c["rc"] and got == c["error"]), ("slow_line", wave, i, c["raw"], got)
            continue
        try:
            cnt, _, _ = orc.filter(c["raw"], want_hits=False)
            got = ("ok", {sv: [int(cnt[j, 0]), int(cnt[j, 1])] for j, sv in enumerate(orc.sv_ids) if cnt[j].sum()})
        except Exception as e:
            got = ("died", type(e).__name__)
        assert got == _want(c), ("C oracle", i, c["raw"])
        for tables, wave in ((True, 0), (False, 0), (True, 1), (True, 2), (True, 3), (True, 4)):
            try:
                cnt, _ = sim.classify(g, c["raw"], tables, wave)
                got = ("ok", {g.sv_ids[j]: [int(cnt[j, 0]), int(cnt[j, 1])] for j in range(g.n_slots) if cnt[j].sum()})
            except Exception as e:
                got = ("died", type(e).__name__)
            assert got == _want(c), ("slow_line", tables, wave, i, c["raw"])
