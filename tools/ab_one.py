"""measurement only (GPU box): mean k_classify_main time of the library SVJG_HIP_LIB names over N launches of the resident configs[2] text
(tools/ab.sh runs the variants under build/ in turn, several rounds, so that box-to-box and minute-to-minute drift cancels)"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi
from svjg.graph import Graph
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cache = f"/dev/shm/svjg_ab_{which}"
os.makedirs(cache, exist_ok=True)
pre = os.path.join(cache, "w")
if which == "long":
    seed = 20260515 + 9
    inf = synth.generate(pre, 0, 20_000, 8, "mixed", seed, write_gaf=False, chrom_style="ucsc")
    mk = lambda: synth.gaf_bytes(inf["tables"], seed, 0, 3_000_000, threads=16, shape="long")
else:
    inf = synth.generate(pre, 0, 100_000, 4, "mixed", 20260517, write_gaf=False)
    mk = lambda: synth.gaf_bytes(inf["tables"], 20260517, 0, 10_000_000, threads=16)
if os.path.exists(pre + ".gaf"):
    gaf = np.fromfile(pre + ".gaf", dtype=np.uint8)
else:
    gaf = mk(); gaf.tofile(pre + ".gaf")
g = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
ctx = capi.Context(0); ctx.load_graph(g); ctx.upload(gaf)
ms = []
for i in range(n + 40):
    ctx.reset_counts(); ctx.classify_resident(base_offset=0, want_hits=False)
    if i >= 40:
        ms.append(ctx.kernel_ms()[0])
print(f"{os.path.basename(os.environ.get('SVJG_HIP_LIB', 'shipped'))} {which}: {np.mean(ms):.4f} ms (median {np.median(ms):.4f}, {n} launches)")
