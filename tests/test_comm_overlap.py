"""svjg/filter.py: classify_sharded creates the RCCL communicators of a multi-GPU run (ncclCommInitAll) — by default in front of the
first upload, with SVJG_COMM_OVERLAP=1 in a thread of its own WHILE the GPUs upload and classify, joined in front of the all-reduce (r05: over
eight ranks the call takes seconds, the order of the whole run; r06: opt-in until it has run on hardware).  Stand-in contexts (the host build of the exact routine, tests/hostsim) and a stand-in communicator call that records when it ran."""
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


def test_by_default_the_communicators_are_there_before_any_gpu_works(case, monkeypatch, tmp_path):
    """r06 (advisor): ncclCommInitAll beside hipMalloc / hipFree / kernel launches on the same devices has never run on hardware, so the
    overlap is opt-in (SVJG_COMM_OVERLAP=1); by default the call is made in the caller's thread before the first upload.  What it took is
    left for pick_devices (rccl_init_s)."""
    g, gaf = case
    monkeypatch.delenv("SVJG_COMM_OVERLAP", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
    _install(monkeypatch, init_seconds=0.1)
    main = threading.get_ident()
    total, recs, data = flt.classify_sharded(g, gaf, want_hits=False, devices=[0, 1, 2])
    one, _ = sim.classify(g, open(gaf, "rb").read())
    assert np.array_equal(total, one.astype(np.uint64))
    ev = {k: [e for e in _Ctx.log if e[0] == k] for k in ("init_begin", "init_end", "classify_begin", "allreduce")}
    assert len(ev["init_begin"]) == 1 and ev["init_begin"][0][1] == main           # the caller's own thread
    assert ev["init_end"][0][2] <= min(e[2] for e in ev["classify_begin"])          # over before any GPU begins
    assert 0.09 < flt.rccl_init_s() < 1.0                                           # measured and remembered
    monkeypatch.setenv("SVJG_RCCL_INIT_S", "2.5")                                   # the environment's word wins
    assert flt.rccl_init_s() == 2.5 and flt.min_bytes_per_device() == int(2.5 * flt.INGEST_BYTES_PER_S)
    _install(monkeypatch, init_seconds=0.05, fail=True)
    with pytest.raises(capi.SvjgError):
        flt.classify_sharded(g, gaf, want_hits=False, devices=[0, 1])
    assert [e[0] for e in _Ctx.log].count("close") == 2 and not any(e[0] in ("allreduce", "classify_begin") for e in _Ctx.log)


def test_how_many_gpus_a_file_is_worth(monkeypatch, tmp_path):
    """pick_devices: as many GPUs as get rccl_init_s() x 5.7 GB/s of text each — with the default (5.6 s for one rank, the only measurement
    there is) BASELINE configs[3]'s 21.6 GB take ONE GPU of eight; a machine that measured 0.5 s would cut it seven ways"""
    monkeypatch.delenv("SVJG_DEVICES", raising=False)
    monkeypatch.delenv("SVJG_RCCL_INIT_S", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))

    class _Lib:
        @staticmethod
        def svjg_device_count():
            return 8
    monkeypatch.setattr(capi, "load_library", lambda: _Lib)
    assert flt.rccl_init_s() == flt.RCCL_INIT_S_DEFAULT and flt.min_bytes_per_device() == flt.MIN_BYTES_PER_DEVICE
    assert flt.pick_devices(21_621_759_346) == [0]
    assert flt.pick_devices(70 << 30) == [0, 1] and flt.pick_devices(1 << 40) == list(range(8))
    flt.note_rccl_init_s(0.5, 8)
    assert flt.rccl_init_s() == 0.5 and flt.pick_devices(21_621_759_346) == list(range(7))
    monkeypatch.setenv("SVJG_DEVICES", "all")
    assert flt.pick_devices(1000) == list(range(8))
    open(flt._rccl_init_file(), "w").write("garbage")
    monkeypatch.delenv("SVJG_DEVICES")
    assert flt.rccl_init_s() == flt.RCCL_INIT_S_DEFAULT


def test_communicators_come_to_be_beside_the_classification(case, monkeypatch, tmp_path):
    g, gaf = case
    monkeypatch.setenv("SVJG_COMM_OVERLAP", "1")
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
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


def test_a_slow_communicator_is_waited_for_and_a_failed_one_raises(case, monkeypatch, tmp_path):
    g, gaf = case
    monkeypatch.setenv("SVJG_COMM_OVERLAP", "1")
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
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
