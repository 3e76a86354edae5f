#!/usr/bin/env python3
"""Exploratory (GPU box): tests/graph_fuzz.py for a range of seeds, HIP path (main kernel, then exact path only) against the C oracle:
counts, and every 10th seed the JSON text against the Python oracle.    python tests/graph_fuzz_campaign.py [first_seed] [n_seeds]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from tests import graph_fuzz
from oracle import oracle_c as OC, oracle_py as O
from svjg import capi
from svjg.graph import Graph

first, n_seeds = (int(sys.argv[1]) if len(sys.argv) > 1 else 5000), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
ctx = capi.Context(0)
tmp = tempfile.mkdtemp()
bad = 0
deferred = lines_total = 0
for seed in range(first, first + n_seeds):
    edges, alt, lines = graph_fuzz.make_case(seed, 2500)
    data = np.frombuffer("".join(lines).encode(), dtype=np.uint8)
    orc = OC.COracle(edges, alt)
    want, _, n = orc.filter(data, want_hits=False)
    wd = {sv: (int(want[i, 0]), int(want[i, 1])) for i, sv in enumerate(orc.sv_ids) if want[i].sum()}
    for all_slow in (False, True):
        g = Graph(edges, alt, all_slow=all_slow)
        ctx.load_graph(g); ctx.reset_counts(); ctx.classify(data, want_hits=(seed % 10 == 0))
        c = ctx.counts()
        got = {g.sv_ids[i]: (int(c[i, 0]), int(c[i, 1])) for i in range(g.n_slots) if c[i].sum()}
        ok = got == wd and ctx.stats()["n_lines"] == n
        if ok and seed % 10 == 0:
            capi.write_informative_json(os.path.join(tmp, "o.json"), data, ctx.hits(), g.sv_ids)
            ok = open(os.path.join(tmp, "o.json")).read() == O.dump_informative(O.classify(lines, edges, alt))
        if not ok:
            bad += 1
            print(f"seed {seed} all_slow={all_slow}: DIFFERENT", flush=True)
        if not all_slow:
            deferred += ctx.stats()["n_deferred"]; lines_total += n
    if (seed - first) % 20 == 19:
        print(f"{seed - first + 1} seeds, {bad} different, {deferred} of {lines_total} lines through the exact path", flush=True)
print(f"graph fuzz: seeds {first}..{first + n_seeds - 1}: {bad} different; {deferred} of {lines_total} lines through the exact path")
sys.exit(1 if bad else 0)
