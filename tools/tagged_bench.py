"""measurement only: the configs[2] text with an id:f:<decimal> tag on every line (GraphAligner style) against the same text without"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "svjedi-graph_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import synth
from svjg import capi
from svjg.graph import Graph
n_aln, n_sv, n_chrom, mix, seed = 4_000_000, 100_000, 4, "mixed", 20260517
tmp = tempfile.mkdtemp(); pre = os.path.join(tmp, "w")
inf = synth.generate(pre, 0, n_sv, n_chrom, mix, seed, write_gaf=False)
gaf = synth.gaf_bytes(inf["tables"], seed, 0, n_aln, threads=16)
graph = Graph.from_files(pre + "_svs_edges.json", pre + ".gfa")
ctx = capi.Context(0); ctx.load_graph(graph)
for name, text in (("plain", gaf), ("id:f:0.9871 on every line", np.frombuffer(bytes(gaf).replace(b"\tdv:f:", b"\tid:f:0.9871\tdv:f:"), dtype=np.uint8)),
                   ("id:f:9e-1 on every line (exact path)", np.frombuffer(bytes(gaf[:gaf.size // 8]).rsplit(b"\n", 1)[0].replace(b"\tdv:f:", b"\tid:f:9e-1\tdv:f:") + b"\n", dtype=np.uint8))):
    ctx.upload(text)
    for it in range(3):
        ctx.reset_counts(); ctx.classify_resident(base_offset=0, want_hits=False)
    st = ctx.stats(); ms = ctx.kernel_ms()
    print(f"{name}: {st['n_lines']} lines, {text.size / 1e6:.0f} MB, deferred {st['n_deferred']}, kernels main {ms[0]:.3f} ms exact {ms[1]:.3f} ms -> {st['n_lines'] / (ms[0] + ms[1]) / 1e6:.1f} G lines/s".replace("G lines/s", "M lines/ms"))
