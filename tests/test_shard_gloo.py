"""Multi-rank path on CPU: byte-range sharding of a GAF at line boundaries + ONE sum all-reduce of the per-SV
count vector (svjedi-graph_amd/svjg/shard.py, tools/dist_boot.py), world_size 2 over gloo.  On the GPU the per-rank counts come from
libsvjg_hip and the all-reduce is RCCL inside the library (svjg_allreduce_counts); here the per-rank counts come
from the CPU oracle so that the sharding / reduction logic is what is under test."""
import json
import os
import socket
import sys

import numpy as np
import pytest

from oracle import oracle_c as OC
from oracle import oracle_py as O
from svjg import shard
import dist_boot


def test_cut_points_keep_lines_whole():
    rng = np.random.default_rng(3)
    lines = []
    for i in range(400):
        body = bytes(rng.integers(97, 123, size=int(rng.integers(1, 200))).astype(np.uint8))
        lines.append(body + (b"\n", b"\r\n", b"\r")[i % 3])
    data = np.frombuffer(b"".join(lines), dtype=np.uint8)
    for n in (1, 2, 3, 8, 64):
        cuts = shard.cut_points(data, n)
        assert cuts[0] == 0 and cuts[-1] == data.size and cuts == sorted(cuts) and len(cuts) == n + 1
        for c in cuts[1:-1]:
            if 0 < c < data.size:
                assert data[c - 1] in (10, 13)                     # right after a terminator
                assert not (data[c - 1] == 13 and data[c] == 10)   # never between \r and \n
        # universal-newline line count is preserved
        text = data.tobytes().decode()
        assert sum(len(data[cuts[r]:cuts[r + 1]].tobytes().decode().splitlines()) for r in range(n)) == len(text.splitlines())
    assert shard.cut_points(np.zeros(0, dtype=np.uint8), 4) == [0, 0, 0, 0, 0]


def _worker(rank, world, port, pre, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
        mine, off = shard.file_shard(pre + ".gaf", world, rank)
        counts, _, n_lines = orc.filter(mine, want_hits=False)
        total = dist_boot.torch_allreduce_counts(counts)
        uid = dist_boot.torch_exchange(bytes(range(128)) if rank == 0 else None)     # the RCCL-id bootstrap path
        np.save(f"{out}.{rank}.npy", total)
        json.dump({"lines": int(n_lines), "offset": int(off), "uid_ok": uid == bytes(range(128))}, open(f"{out}.{rank}.json", "w"))
    finally:
        dist.destroy_process_group()


def test_two_rank_allreduce_equals_whole_file(tmp_path):
    import synth
    import torch.multiprocessing as mp
    pre = str(tmp_path / "s")
    synth.generate(pre, 20000, 400, 2, "mixed", 11)
    orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
    whole, _, n_lines = orc.filter(np.fromfile(pre + ".gaf", dtype=np.uint8), want_hits=False)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(2, port, pre, out), nprocs=2, join=True)
    r0, r1 = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    assert np.array_equal(r0, whole) and np.array_equal(r1, whole)
    m0, m1 = json.load(open(out + ".0.json")), json.load(open(out + ".1.json"))
    assert m0["lines"] + m1["lines"] == n_lines and m0["offset"] == 0 and m1["offset"] > 0
    assert m0["uid_ok"] and m1["uid_ok"]
