#!/usr/bin/env python3
"""Exploratory (GPU box): tests/longpath_fuzz.py for a range of seeds — graphs of >= 2 000 nodes, walks of 65..216 nodes with ONE late event —
HIP path (main kernel, then exact path only) against the C oracle: counts, with hit records on every 5th seed and then the JSON text against
the Python oracle; every seed's fatal lines (an insertion node the GFA lacks) must die with the oracle's exception.
    python tests/longpath_fuzz_campaign.py [first_seed] [n_seeds] [long lines per seed]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from tests import longpath_fuzz                      # noqa: E402
from oracle import oracle_c as OC, oracle_py as O    # noqa: E402
from svjg import capi                                # noqa: E402
from svjg.graph import Graph                         # noqa: E402

first, n_seeds = (int(sys.argv[1]) if len(sys.argv) > 1 else 30000), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
n_long = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ctx = capi.Context(0)
tmp = tempfile.mkdtemp()
bad = deferred = lines_total = fatal_ok = fatal_n = 0
events = {}
for seed in range(first, first + n_seeds):
    edges, alt, lines, fatal = longpath_fuzz.make_case(seed, n_long, 4)
    for l in lines:
        nm = l.split("\t", 1)[0]
        if "_k" in nm:
            ev = nm.split("_", 2)[2].rsplit("_k", 1)[0]
            events[ev] = events.get(ev, 0) + 1
    data = np.frombuffer("".join(lines).encode(), dtype=np.uint8)
    orc = OC.COracle(edges, alt)
    want, _, n = orc.filter(data, want_hits=False)
    wd = {sv: (int(want[i, 0]), int(want[i, 1])) for i, sv in enumerate(orc.sv_ids) if want[i].sum()}
    for all_slow in (False, True):
        g = Graph(edges, alt, all_slow=all_slow)
        ctx.load_graph(g); ctx.reset_counts(); ctx.classify(data, want_hits=(seed % 5 == 0))
        c = ctx.counts()
        got = {g.sv_ids[i]: (int(c[i, 0]), int(c[i, 1])) for i in range(g.n_slots) if c[i].sum()}
        ok = got == wd and ctx.stats()["n_lines"] == n
        if ok and seed % 5 == 0:
            capi.write_informative_json(os.path.join(tmp, "o.json"), data, ctx.hits(), g.sv_ids)
            ok = open(os.path.join(tmp, "o.json")).read() == O.dump_informative(O.classify(lines, edges, alt))
        if not ok:
            bad += 1
            print(f"seed {seed} all_slow={all_slow}: DIFFERENT", flush=True)
        if not all_slow:
            deferred += ctx.stats()["n_deferred"]; lines_total += n
            for f in fatal:
                raw = "".join(lines[:5] + [f] + lines[5:7]).encode()
                e1 = e2 = None
                try:
                    orc.filter(raw, want_hits=False)
                except Exception as e:                # noqa: BLE001
                    e1 = type(e).__name__
                try:
                    ctx.reset_counts(); ctx.classify(np.frombuffer(raw, dtype=np.uint8), want_hits=True)
                except Exception as e:                # noqa: BLE001
                    e2 = type(e).__name__
                fatal_n += 1
                fatal_ok += e1 == e2 and e1 is not None
    if (seed - first) % 20 == 19:
        print(f"{seed - first + 1} seeds, {bad} different, {deferred} of {lines_total} lines through the exact path, fatal lines {fatal_ok} of {fatal_n} with the oracle's exception", flush=True)
print(f"long-path fuzz: seeds {first}..{first + n_seeds - 1}: {bad} different; {deferred} of {lines_total} lines through the exact path; fatal lines {fatal_ok} of {fatal_n} with the oracle's exception")
print("late events:", dict(sorted(events.items())))
sys.exit(1 if bad or fatal_ok != fatal_n else 0)
