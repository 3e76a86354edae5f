import os, sys, time, tempfile
import numpy as np
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi, genotype
from svjg.graph import Graph
n_aln, n_sv, n_chrom, mix, seed = 10_000_000, 100_000, 4, "mixed", 20260517
tmp = tempfile.mkdtemp(); pre = os.path.join(tmp, "w")
inf = synth.generate(pre, 0, n_sv, n_chrom, mix, seed, write_gaf=False)
gaf = synth.gaf_bytes(inf["tables"], seed, 0, n_aln, threads=16)
graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
rows = genotype.VcfRows(pre + ".vcf", graph.slot_of)
ctx = capi.Context(0); ctx.load_graph(graph); ctx.upload(gaf)
T = {"reset":0,"classify":0,"geno":0}
for it in range(12):
    ctx.sync()
    t0=time.perf_counter(); ctx.reset_counts(); t1=time.perf_counter()
    ctx.classify_resident(base_offset=0, want_hits=False); t2=time.perf_counter()
    ctx.genotype(rows.sv_type, rows.slot, rows.ok, 3, 0.00005, reuse_outputs=True); t3=time.perf_counter()
    if it>=2:
        T["reset"]+=t1-t0; T["classify"]+=t2-t1; T["geno"]+=t3-t2
print({k: round(v/10*1e3,3) for k,v in T.items()}, ctx.kernel_ms())
