import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.environ["GRAFT_REPO_ROOT"]
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from svjg import capi, filter as flt
from svjg.graph import Graph
r = f"{ROOT}/tests/golden/realshape"
ctx = capi.Context(0)
g = Graph.from_files(f"{r}/r_svs_edges.json", f"{r}/r.gfa")
counts, recs, data = flt.classify_file(ctx, g, f"{r}/r.gaf")
print(ctx.stats(), ctx.defer_causes())
