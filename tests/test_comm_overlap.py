"""svjg/filter.py: classify_sharded creates the RCCL communicators of a multi-GPU run (ncclCommInitAll) in a thread of its own WHILE the
GPUs upload and classify, and joins it in front of the all-reduce (r05: over eight ranks the call takes seconds, the order of the whole
run).  Stand-in contexts (the host build of the exact routine, tests/hostsim) and a stand-in communicator call that records when it ran."""
import os
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

from svjg import capi, filter as flt          # noqa: E402
from svjg.graph import Graph                   # noqa: E402
from tests.hostsim import sim                  # noqa: E402


class _Ctx:
    log = []

    def __init__(self, device):
        self.device, self.total, self.comm = device, None, False

    def load_graph(self, g):
        self.g = g
        self.total = np.zeros((g.n_slots, 2), dtype=np.uint64)

    def classify_file(self, path, off, n, want_hits=False):
        _Ctx.log.append(("classify_begin", self.device, time.perf_counter()))
        with open(path, "rb") as fh:
            fh.seek(off)
            raw = fh.read(n)
        time.sleep(0.15)                                     # (the upload and the kernels)
        c, _ = sim.classify(self.g, raw)
        self.total += c.astype(np.uint64)
        _Ctx.log.append(("classify_end", self.device, time.perf_counter()))

    def stats(self):
        return {"non_ascii": 0, "n_deferred": 0}

    def host_lines(self):
        return np.zeros(0, dtype=np.uint64)

    def counts(self):
        return self.total

    def close(self):
        _Ctx.log.append(("close", self.device, time.perf_counter()))


def _install(monkeypatch, init_seconds, fail=False):
    _Ctx.log = []

    def comm_init_all(ctxs):
        _Ctx.log.append(("init_begin", threading.get_ident(), time.perf_counter()))
        time.sleep(init_seconds)
        if fail:
            _Ctx.log.append(("init_end", threading.get_ident(), time.perf_counter()))
            raise capi.SvjgError("ncclCommInitAll: unhandled system error")
        for c in ctxs:
            c.comm = True
        _Ctx.log.append(("init_end", threading.get_ident(), time.perf_counter()))

    def allreduce_counts_all(ctxs):
        assert len(ctxs) == 1 or all(c.comm for c in ctxs)    # the communicators are there when the collective is issued
        _Ctx.log.append(("allreduce", 0, time.perf_counter()))
        tot = sum(c.total for c in ctxs)
        for c in ctxs:
            c.total = tot.copy()
    monkeypatch.setattr(capi, "Context", _Ctx)
    monkeypatch.setattr(capi, "comm_init_all", comm_init_all)
    monkeypatch.setattr(capi, "allreduce_counts_all", allreduce_counts_all)
    monkeypatch.setattr(capi, "release_host_tables", lambda: None)
    monkeypatch.setattr(flt, "resolve_host_lines", lambda ctxs, data, want_hits, err: None)


@pytest.fixture()
def case(golden):
    t = f"{golden}/testdir"
    return Graph.from_files(f"{t}/test_svs_edges.json", f"{t}/test.gfa", native=False), f"{t}/test.gaf"


def test_communicators_come_to_be_beside_the_classification(case, monkeypatch):
    g, gaf = case
    _install(monkeypatch, init_seconds=0.1)
    main = threading.get_ident()
    total, recs, data = flt.classify_sharded(g, gaf, want_hits=False, devices=[0, 1, 2])
    one, _ = sim.classify(g, open(gaf, "rb").read())
    assert np.array_equal(total, one.astype(np.uint64))        # three shards summed = the file
    ev = {k: [e for e in _Ctx.log if e[0] == k] for k in ("init_begin", "init_end", "classify_begin", "classify_end", "allreduce", "close")}
    assert len(ev["init_begin"]) == 1 and ev["init_begin"][0][1] != main          # a thread of its own
    t_init0, t_init1 = ev["init_begin"][0][2], ev["init_end"][0][2]
    assert t_init0 < min(e[2] for e in ev["classify_end"])                      # started before any GPU was done ...
    assert t_init1 < max(e[2] for e in ev["classify_end"])                      # ... and over while they still worked: nothing waited for it
    assert t_init1 <= ev["allreduce"][0][2] < min(e[2] for e in ev["close"])    # joined in front of the collective


def test_a_slow_communicator_is_waited_for_and_a_failed_one_raises(case, monkeypatch):
    g, gaf = case
    _install(monkeypatch, init_seconds=0.6)                      # longer than the classification: the all-reduce waits for it
    flt.classify_sharded(g, gaf, want_hits=False, devices=[0, 1])
    ev = {k: [e[2] for e in _Ctx.log if e[0] == k] for k in ("init_end", "classify_end", "allreduce")}
    assert max(ev["classify_end"]) < ev["init_end"][0] <= ev["allreduce"][0]
    _install(monkeypatch, init_seconds=0.05, fail=True)
    with pytest.raises(capi.SvjgError):
        flt.classify_sharded(g, gaf, want_hits=False, devices=[0, 1])
    assert [e[0] for e in _Ctx.log].count("close") == 2 and not any(e[0] == "allreduce" for e in _Ctx.log)
    # one GPU: no communicator at all
    _install(monkeypatch, init_seconds=0.05)
    flt.classify_sharded(g, gaf, want_hits=False, devices=[0])
    assert not any(e[0] == "init_begin" for e in _Ctx.log)
