#!/usr/bin/env python3
"""Exploratory differential fuzz on the GPU box: thousands of mutated lines (mutations of tests/golden/make_fuzz.py) mixed into a
synthetic batch, HIP path against the C oracle (itself pinned on the reference by golden/fuzz).  Lines the oracle dies on
are left out of the mixed file and tried one by one for the exception class.

    python tests/fuzz_big.py [n_mutants] [seed]
"""
import os, random, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import synth
from make_fuzz import mutate, MORE_TAGS
from oracle import oracle_c as OC, oracle_py as O
from svjg import capi
from svjg.graph import Graph

n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp(); pre = os.path.join(tmp, "f")
inf = synth.generate(pre, 60000, 1500, 3, "mixed", 4242, write_gaf=False, return_gaf=True)
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
lines = inf["gaf"].tobytes().splitlines(keepends=True)
ctx = capi.Context(0); ctx.load_graph(g)

def cd(counts):
    return {g.sv_ids[i]: (int(counts[i, 0]), int(counts[i, 1])) for i in range(g.n_slots) if counts[i].sum()}
def od(want):
    return {sv: (int(want[i, 0]), int(want[i, 1])) for i, sv in enumerate(orc.sv_ids) if want[i].sum()}

good, bad = [], []
for _ in range(n_mut):
    m = mutate(rng.choice(lines), rng, MORE_TAGS)
    if rng.random() < 0.02:                        # r04: a tail that runs past the main kernel's staged 8 KB (plain, or with what sends the line to the exact path)
        body = m.rstrip(b"\r\n")
        fill = rng.choice((b"12M3D", b"ACGT", b"A", b"7M", b"9I2D", b"d", b"5M:d"))
        extra = rng.choice((b"", b"", b"", b"\tid:f:0.9", b"\tid:f:0", b"\rx", b"\txd:i:1", b" ", b"\t"))
        m = body + b"\tcg:Z:" + fill * (rng.choice((7000, 8200, 12300, 16380, 24000, 31000)) // len(fill)) + extra + m[len(body):]
    if b"\xd9\xa3" in m or any(c >= 0x80 for c in m):
        continue                                   # documented divergences / UTF-8 handling is the host's
    if len(m) > 32768:
        continue                                   # (paths of thousands of nodes are O(k^2) in the oracle: they take minutes, not seconds)
    try:
        orc.filter(m, want_hits=False)
        good.append(m if m.endswith((b"\n", b"\r")) else m + b"\n")
    except OC.Undecided:
        continue                                   # (a number beyond what the C restatement holds: the Python oracle's business, tests/test_oracle_cross_fuzz.py)
    except Exception as e:
        bad.append((m, type(e).__name__))
mixed = []
gi = iter(good)
for i, l in enumerate(lines):
    mixed.append(l)
    if i % 7 == 0:
        x = next(gi, None)
        if x is not None:
            mixed.append(x)
mixed += list(gi)
data = b"".join(mixed)
want, _, n_lines = orc.filter(data, want_hits=False)
ctx.reset_counts(); ctx.classify(np.frombuffer(data, dtype=np.uint8))
st = ctx.stats()
ok = cd(ctx.counts()) == od(want) and st["n_lines"] == n_lines
print(f"mixed file: {len(good)} accepted mutants among {len(lines)} lines, {st['n_deferred']} lines through the exact path: {'EQUAL' if ok else 'DIFFERENT'}", flush=True)
n_bad_ok = 0
pad = b"".join(lines[:30])
for m, cls in bad[:600]:
    ctx.reset_counts()
    try:
        ctx.classify(np.frombuffer(pad + m + (b"" if m.endswith((b"\n", b"\r")) else b"\n") + pad, dtype=np.uint8))
        got = "accepted"
    except Exception as e:
        got = type(e).__name__
    if got == cls:
        n_bad_ok += 1
    else:
        print("MISMATCH", cls, got, m[:200], flush=True)
print(f"fatal mutants: {n_bad_ok} of {min(len(bad), 600)} with the oracle's exception class")
sys.exit(0 if ok and n_bad_ok == min(len(bad), 600) else 1)
