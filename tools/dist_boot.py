"""Launcher-side helpers for the one-process-per-GPU runs (bench.py under torch.distributed.run, the gloo tests): the
bootstrap that carries RCCL's 128-byte unique id from rank 0 to the other ranks, and a host-side sum used by the CPU
tests of the sharding logic.  Not part of the product package: libsvjg_hip's own collective is RCCL (svjg_allreduce_counts)."""
import numpy as np


def torch_exchange(uid):
    """Broadcast rank 0's unique id through torch.distributed (any backend that moves CPU bytes)."""
    import torch
    import torch.distributed as dist
    t = torch.zeros(128, dtype=torch.uint8)
    if uid is not None:
        t = torch.frombuffer(bytearray(uid), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    return bytes(t.numpy().tobytes())


def torch_allreduce_counts(counts):
    """Host-side sum of a count vector over torch.distributed (gloo)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(counts.astype(np.int64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy().astype(counts.dtype)
