"""Stand-in for svjg.capi in CPU tests of bench.py's launcher plumbing (SVJG_BENCH_CAPI=tests.standin_capi): a context classifies
its resident shard with the host build of the exact per-line routine (tests/hostsim) and sums its count vector over
torch.distributed (gloo) where the library would call RCCL.  Test infrastructure only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.hostsim import sim          # noqa: E402


def device_count():
    return int(os.environ.get("SVJG_STANDIN_DEVICES", "8"))


def unique_id():
    return bytes(range(128))


def release_host_tables():
    pass


class RcclGroup:
    def __init__(self, ctx, n_ranks, rank, exchange):
        uid = exchange(unique_id() if rank == 0 else None)
        assert uid == unique_id()                                  # the id of rank 0 reached this rank
        ctx.world = n_ranks


class Context:
    def __init__(self, device=0):
        self.device, self.world, self.pending, self.n_lines = device, 1, 0, 0

    def load_graph(self, graph):
        self.graph = graph

    def set_rows(self, sv_type, slot, ok):
        self.n_rows = len(sv_type)

    def upload(self, text):
        self.text = np.array(text, copy=True)

    def run_begin(self, min_support, err):
        self.pending += 1
        assert self.pending <= 2

    def run_end(self):
        self.pending -= 1
        counts, self.n_lines = sim.classify(self.graph, self.text)
        if self.world > 1:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(counts.astype(np.int64))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            counts = t.numpy()
        self.total = counts.astype(np.uint64)
        n = self.n_rows
        return np.zeros(n, np.uint8), np.zeros((n, 3), np.int32), np.zeros((n, 2), np.uint32), np.ones(n, np.uint8)

    def kernel_ms(self):
        return 1.0, 0.0, 0.1

    def allreduce_on_second_stream(self, on):                      # (bench.py's second leg at N > 1: a flag here)
        self.second = bool(on)

    def sync(self):
        pass

    def close(self):
        pass

    def stats(self):
        return {"n_lines": self.n_lines, "n_deferred": 0, "n_hitrecs": 0, "non_ascii": 0}

    def counts(self):
        return self.total


def comm_init_all(ctxs):
    raise RuntimeError("the stand-in has no single-process communicator (tests/test_bench_local_ranks.py covers that form)")
