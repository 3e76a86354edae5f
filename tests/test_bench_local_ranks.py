"""bench.py's single-process multi-GPU driver (one thread per context, the contexts' all-reduce keeps the threads in step) on
the CPU: the contexts are stand-ins that classify their shard with the host build of the exact per-line routine (tests/hostsim)
and sum their count vectors the way svjg_run_resident's RCCL leg does — so the threading, the barriers, the per-rank shards of
the synthetic stream and the error path are exercised without a GPU.  (The real thing: tests/test_gpu_parity.py
test_bench_single_process_two_gpus, test_run_resident_is_the_three_calls.)"""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench          # noqa: E402
import synth          # noqa: E402
from svjg.graph import Graph      # noqa: E402
from tests.hostsim import sim     # noqa: E402


class _Group:
    """what ncclAllReduce does for the stand-ins: every rank adds its vector, all leave with the sum"""

    def __init__(self, n):
        self.n, self.bar, self.acc, self.lock = n, threading.Barrier(n), None, threading.Lock()

    def allreduce(self, v):
        with self.lock:
            self.acc = v.astype(np.int64) if self.acc is None else self.acc + v
        self.bar.wait()
        out = self.acc.copy()
        if self.bar.wait() == 0:
            self.acc = None
        self.bar.wait()
        return out


class _StandIn:
    def __init__(self, graph, text, group, fail_at=None):
        self.graph, self.text, self.group, self.calls, self.fail_at = graph, text, group, 0, fail_at

    def run_begin(self, min_support, err):
        self.pending = getattr(self, "pending", 0) + 1
        assert self.pending <= 2                                   # (the library takes two passes in flight)

    def run_end(self):
        self.pending -= 1
        return self.run_resident(3, 0.00005)

    def run_resident(self, min_support, err):
        self.calls += 1
        if self.fail_at is not None and self.calls == self.fail_at:
            raise ValueError("malformed GAF line")
        counts, n_lines = sim.classify(self.graph, self.text)
        total = self.group.allreduce(counts) if self.group else counts
        raw = total.astype(np.uint32)
        return np.zeros(len(raw), np.uint8), np.zeros((len(raw), 3), np.int32), raw, (raw.sum(1) > 0).astype(np.uint8)

    def kernel_ms(self):
        return 1.0, 0.0, 0.1

    def sync(self):
        pass


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    pre = str(tmp_path_factory.mktemp("b") / "w")
    inf = synth.generate(pre, 0, 400, 2, "mixed", 5, write_gaf=False)
    g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
    shards = [synth.gaf_bytes(inf["tables"], 5, r * 3000, 3000, threads=2) for r in range(3)]
    return g, shards


@pytest.mark.parametrize("n", [1, 3])
def test_timed_steps_threads_and_shards(case, n):
    g, shards = case
    grp = _Group(n) if n > 1 else None
    ctxs = [_StandIn(g, shards[r], grp) for r in range(n)]
    outer = []
    dt, ms, out = bench.timed_steps(ctxs, steps=3, warmup=2, outer_barrier=lambda: outer.append(1))
    assert dt > 0 and len(outer) == 2 and [len(m) for m in ms] == [3] * n and all(c.calls == 5 for c in ctxs)
    whole, n_lines = sim.classify(g, np.concatenate(shards[:n]))
    assert n_lines == 3000 * n and np.array_equal(out[2], whole) and whole.sum() > 0
    # rank r's shard is lines [r * 3000, (r + 1) * 3000) of ONE stream: the shards differ and their line counts add up
    assert len({s.tobytes() for s in shards[:n]}) == n


def test_timed_steps_reports_a_rank_that_fails(case):
    g, shards = case
    grp = _Group(2)
    ctxs = [_StandIn(g, shards[0], grp), _StandIn(g, shards[1], grp, fail_at=1)]
    grp.bar = threading.Barrier(2, timeout=5)                   # (the surviving rank must not wait for ever)
    with pytest.raises((ValueError, threading.BrokenBarrierError)):
        bench.timed_steps(ctxs, steps=2, warmup=1)


def test_bench_refuses_more_gpus_than_there_are(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher on a box with fewer devices: non-zero exit with the count, no JSON line"""
    from svjg import capi
    monkeypatch.setattr(capi, "device_count", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--no-cpu-baseline", "--no-e2e"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert "needs 2 devices, found 1" in str(ei.value.code) and not capsys.readouterr().out.strip()


def test_bench_under_the_launcher_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 …` — the form the scaling runs use — with the library
    replaced by tests/standin_capi.py: rendezvous on 127.0.0.1, the unique id travelling from rank 0, one shard of the synthetic
    stream per rank, the barriers around the timed passes, the maximum over the ranks, ONE JSON line from rank 0."""
    import json
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SVJG_BENCH_CAPI="tests.standin_capi", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--aln", "2000", "--svs", "300", "--no-e2e", "--no-cpu-baseline", "--north-star-aln", "3000", "--north-star-svs", "200"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["rccl"]["ranks"] == 2 and r["config"]["alignments_per_gpu"] == 2000
    assert abs(r["value"] - 2 * 2000 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-6 * r["value"]
    # the timed passes again until they span a quarter of a second, and the north_star workload split over the two ranks
    sus = r["sustained"]
    # (the loop is sized from the two timed steps to span 0.25 s; on a busy host those two can come out slow and the loop then falls short of it)
    assert sus["steps"] >= 2 and sus["seconds"] >= 0.05 and abs(sus["alignments_per_s"] - 2 * 2000 * sus["steps"] / sus["seconds"]) < 1e-6 * sus["alignments_per_s"]
    ns = r["north_star"]
    assert ns["n_gpus"] == 2 and ns["alignments"] == 3000 and ns["alignments_per_gpu"] == 1500 and ns["passes"] == 3
    assert ns["digest_equal_across_ranks"] is True and len(ns["counts_digest"]) == 16 and ns["alignments_per_s"] > 0
    assert "allreduce_stream" in r["rccl"] and "debug_info" in r["rccl"]
    ars = r["rccl"]["allreduce_stream"]                           # both placements of the pass's all-reduce were timed
    assert ars["compute"]["ms_per_step"] > 0 and ars["second"]["steps"] >= 2 and ars["second"]["ms_per_step"] > 0
