"""measurement only: where a worker's time goes, phase by phase (needs build/lib_timing.so: tools/mkvariant.sh timing -DSVJG_ABLATE -DSVJG_TIMING;
run on the GPU box with SVJG_DIAG=16: the library prints the cycle sums of lane 0 of every worker)"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SVJG_HIP_LIB"] = os.path.join(ROOT, "build", "lib_timing.so")      # (selected, not copied over the shipped library)
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import synth
from svjg import capi
from svjg.graph import Graph
tmp = tempfile.mkdtemp(); pre = os.path.join(tmp, "w")
if len(sys.argv) > 1 and sys.argv[1] == "hg002":                 # bench.py's hg002_shape block
    inf = synth.generate_hg002(pre, n_reads=synth.HG002_READS, write_gaf=False, return_gaf=True, threads=16)
    gaf = inf["gaf"]
elif len(sys.argv) > 1 and sys.argv[1] == "long":                # bench.py's long_read block
    seed = 20260515 + 9
    inf = synth.generate(pre, 0, 20_000, 8, "mixed", seed, write_gaf=False, chrom_style="ucsc")
    gaf = synth.gaf_bytes(inf["tables"], seed, 0, 3_000_000, threads=16, shape="long")
else:
    n_aln, n_sv, n_chrom, mix, seed = 10_000_000, 100_000, 4, "mixed", 20260517
    inf = synth.generate(pre, 0, n_sv, n_chrom, mix, seed, write_gaf=False)
    gaf = synth.gaf_bytes(inf["tables"], seed, 0, n_aln, threads=16)
graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
ctx = capi.Context(0); ctx.load_graph(graph); ctx.upload(gaf)
for it in range(3):
    ctx.reset_counts(); ctx.classify_resident(base_offset=0, want_hits=False)
print(ctx.kernel_ms())
