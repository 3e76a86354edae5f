"""The fused pass (svjg_run_begin / svjg_run_end) on several ranks, on the CPU: a model of the ranks' call sequence — threads in
place of GPUs, a barrier-and-sum in place of ncclAllReduce (which, like RCCL, pairs the ranks' collectives BY ORDER OF ISSUE and
hangs when a rank issues one the others do not), the host build of the exact per-line routine in place of the kernels — driven by
the SAME decision code libsvjg_hip.so uses (svjedi-graph_amd/csrc/svjg_pass.h through tests/hostsim).  A rank whose list of
deferred lines overflows makes EVERY rank repeat the pass; the round-3 form (each rank decides by itself) is kept as a negative
control: the model must see it hang or mis-sum.  (The real thing on a GPU: tests/test_gpu_parity.py
test_fused_pass_with_many_deferred_lines_under_a_communicator.)"""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth                      # noqa: E402
from svjg.graph import Graph      # noqa: E402
from tests.hostsim import sim     # noqa: E402


class Collective:
    """sum all-reduce over n ranks; the k-th call of every rank pairs with the k-th call of the others; a rank left alone times out"""

    def __init__(self, n, timeout=5.0):
        self.n, self.bar, self.acc, self.lock = n, threading.Barrier(n, timeout=timeout), None, threading.Lock()

    def allreduce(self, v):
        with self.lock:
            self.acc = v.astype(np.int64) if self.acc is None else self.acc + v
        self.bar.wait()
        out = self.acc.copy()
        if self.bar.wait() == 0:
            self.acc = None
        self.bar.wait()
        return out


class ModelRank:
    """one context: a shard, a list of deferred lines of `cap` entries.  A line holding b"id:f:" is 'deferred'; when more of them than
    the list holds turn up, the pass's counts lack them and the overflow bit is set (what k_classify_main does)."""

    def __init__(self, graph, text, comm, cap, collective_decision=True):
        self.g, self.text, self.comm, self.cap, self.collective = graph, text, comm, cap, collective_decision
        self.repeat_word, self.repeats, self.counts_overflowed, self.gw = sim.pass_logic()
        self.n_allreduce = 0

    def _classify(self, complete):
        lines = bytes(self.text).splitlines(True)
        deferred = [l for l in lines if b"id:f:" in l]
        overflow = 1 if (len(deferred) > self.cap and not complete) else 0
        keep = lines if not overflow else [l for l in lines if b"id:f:" not in l]
        counts, _ = sim.classify(self.g, np.frombuffer(b"".join(keep), dtype=np.uint8)) if keep else (np.zeros((self.g.n_slots, 2), np.uint32), 0)
        return counts.astype(np.int64), overflow

    def _reduce(self, vec):
        self.n_allreduce += 1
        return self.comm.allreduce(vec) if self.comm else vec

    def run_pass(self):
        counts, overflow = self._classify(complete=False)                      # svjg_run_begin: kernels + guard kernel + the pass's all-reduce
        flat = counts.reshape(-1)
        guard = np.array([flat[0::2].max(initial=0), flat[1::2].max(initial=0), self.repeat_word(overflow)], np.int64)
        assert len(guard) == self.gw
        red = self._reduce(np.concatenate([flat, guard])) if self.comm else np.concatenate([flat, guard])
        g = red[-self.gw:]
        repeat = self.repeats(self.comm is not None, overflow, int(g[2])) if self.collective else bool(overflow)   # svjg_run_end
        if repeat:
            counts, _ = self._classify(complete=True)                           # classify_range: sizes the list, retries
            red = self._reduce(np.concatenate([counts.reshape(-1), np.zeros(self.gw, np.int64)])) if self.comm else counts.reshape(-1)
        assert not self.counts_overflowed(int(g[0]), int(g[1]))
        return red[: counts.size].reshape(-1, 2)


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    pre = str(tmp_path_factory.mktemp("p") / "w")
    inf = synth.generate(pre, 0, 300, 2, "mixed", 11, write_gaf=False)
    g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    shards = [synth.gaf_bytes(inf["tables"], 11, r * 1500, 1500, threads=2) for r in range(2)]
    # rank 1's shard: an exponent-form identity tag on every line (the main kernel leaves those to the exact path)
    shards[1] = np.frombuffer(bytes(shards[1]).replace(b"\tdv:f:", b"\tid:f:5e-1\tdv:f:"), dtype=np.uint8)
    whole, n = sim.classify(g, np.concatenate(shards))
    assert n == 3000 and whole.sum() > 0
    return g, shards, whole


def _run(ranks, passes=2):
    out, errs = [None] * len(ranks), []

    def work(i):
        try:
            for _ in range(passes):
                out[i] = ranks[i].run_pass()
        except BaseException as e:          # noqa: BLE001
            errs.append(e)
            ranks[i].comm.bar.abort()
    th = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(len(ranks))]
    for t in th:
        t.start()
    for t in th:
        t.join(30)
    return out, errs


def test_decision_function():
    word, repeats, over, gw = sim.pass_logic()
    assert gw == 3 and word(0) == 0 and word(1) == 1 and word(5) == 1
    assert repeats(False, 1, 0) and not repeats(False, 0, 7)              # alone: the rank's own status
    assert repeats(True, 0, 1) and not repeats(True, 0, 0) and repeats(True, 1, 1)   # under a communicator: the sum over the ranks
    assert not over((1 << 32) - 1, 5) and over(1 << 32, 0) and over(0, 1 << 32)


@pytest.mark.parametrize("caps", [(100, 100), (10000, 100), (10000, 10000)])
def test_one_rank_overflows_all_ranks_repeat(case, caps):
    """exactly one rank overflows (its list holds 100 lines, 1500 are deferred), both, or none: every rank finishes, every rank
    holds the whole file's counts, and all ranks issued the same number of collectives"""
    g, shards, whole = case
    comm = Collective(2)
    ranks = [ModelRank(g, shards[r], comm, caps[r]) for r in range(2)]
    out, errs = _run(ranks)
    assert not errs, errs
    for o in out:
        assert np.array_equal(o, whole)
    overflowed = caps[1] < 1500                                             # (rank 0's shard defers nothing)
    assert ranks[0].n_allreduce == ranks[1].n_allreduce == (4 if overflowed else 2)


def test_rank_local_decision_is_what_the_model_catches(case):
    """negative control — round 3's form: the rank that overflowed repeats by itself and issues an all-reduce its peer never does"""
    g, shards, whole = case
    comm = Collective(2, timeout=1.0)
    ranks = [ModelRank(g, shards[r], comm, (10000, 100)[r], collective_decision=False) for r in range(2)]
    out, errs = _run(ranks, passes=1)
    assert errs or not all(o is not None and np.array_equal(o, whole) for o in out)


def test_single_rank_without_a_communicator(case):
    g, shards, _ = case
    r = ModelRank(g, shards[1], None, 100)
    want, _ = sim.classify(g, shards[1])
    assert np.array_equal(r.run_pass(), want) and r.n_allreduce == 0
