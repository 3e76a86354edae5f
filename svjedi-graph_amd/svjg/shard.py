"""Multi-GPU: one process per GPU, the GAF split into contiguous byte ranges cut at line boundaries, graph
tables replicated, and ONE all-reduce (sum) of the per-SV count vector — over RCCL/xGMI inside libsvjg_hip
(svjg_allreduce_counts).  The genotype pass then runs on the reduced vector.

The reference has nothing distributed (single process, single thread); this module is the design of
SURVEY.md §8(e).  Integer sums make the result independent of the reduction order: bit-exact.
"""
import numpy as np

_TERM = (10, 13)


def cut_points(data, n_ranks):
    """n_ranks + 1 offsets into `data` (uint8 array); shard r = data[cut[r]:cut[r+1]].

    Every cut sits right after a line terminator (\\n, \\r\\n kept together, or a lone \\r), so each line
    belongs to exactly one shard and file order is preserved by rank order."""
    n = int(data.size)
    cuts = [0]
    for r in range(1, n_ranks):
        p = max(cuts[-1], (n * r) // n_ranks)
        while p < n and data[p] not in _TERM:
            # scan forward in blocks
            blk = data[p:p + 65536]
            hit = np.flatnonzero((blk == 10) | (blk == 13))
            if hit.size:
                p += int(hit[0])
                break
            p += blk.size
        if p < n:
            if data[p] == 13 and p + 1 < n and data[p + 1] == 10:
                p += 1
            p += 1
        cuts.append(min(p, n))
    cuts.append(n)
    return cuts


def shard_of(data, n_ranks, rank):
    c = cut_points(data, n_ranks)
    return data[c[rank]:c[rank + 1]], c[rank]


def file_shard(path, n_ranks, rank):
    """Read only this rank's byte range of a GAF file (plus what is needed to find its boundaries)."""
    import os
    n = os.path.getsize(path)
    if n == 0:
        return np.zeros(0, dtype=np.uint8), 0
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    c = cut_points(mm, n_ranks)
    return np.array(mm[c[rank]:c[rank + 1]]), c[rank]


class RcclGroup:
    """RCCL communicator living inside libsvjg_hip; the 128-byte unique id travels over whatever
    bootstrap the launcher offers (`exchange` = callable(bytes|None) -> bytes; bench.py passes a broadcast over its
    launcher's process group, tools/dist_boot.py).  One process per GPU; the single-process form the drop-in scripts use
    is capi.comm_init_all / capi.allreduce_counts_all."""

    def __init__(self, ctx, n_ranks, rank, exchange):
        from . import capi
        uid = capi.unique_id() if rank == 0 else None
        uid = exchange(uid)
        ctx.comm_init(uid, n_ranks, rank)
        self.ctx = ctx

    def allreduce_counts(self):
        self.ctx.allreduce_counts()
