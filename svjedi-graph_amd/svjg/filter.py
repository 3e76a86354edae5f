"""Host side of the alignment filter: the part of filter-alignments.py that is file handling.

    GAF bytes --svjg_classify (HIP)--> per-SV (ref, alt) counts + hit records
              --svjg_write_informative_json (native, host)--> <prefix>_informative_aln.json

The classification itself (filter-alignments.py:123-166) happens in libsvjg_hip.so and the JSON text is produced by
libsvjg_host.so; nothing in this module looks inside an alignment line.
"""
import os

import numpy as np

from . import capi
from .graph import Graph


def output_names(prefix, output_dir=None):
    """filter-alignments.py:75-84."""
    if prefix:
        out = prefix + "_informative_aln" + ".json"
        edges = prefix + "_svs_edges.json"
    else:
        # the reference never assigns svs_edges_dict on this branch and dies with UnboundLocalError (exit 1)
        raise UnboundLocalError("local variable 'svs_edges_dict' referenced before assignment")
    if output_dir:
        out = "/".join([output_dir, out])
    return out, edges


def read_gaf(path):
    """Whole file as uint8 (the reference reads it in text mode; UTF-8 validity is checked later only if a
    non-ASCII byte was seen on the device)."""
    if os.path.getsize(path) == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.fromfile(path, dtype=np.uint8)


def classify_file(ctx, graph, gaf_path, want_hits=True):
    """-> (counts[n_slots, 2], hit records, raw bytes)"""
    data = read_gaf(gaf_path)
    ctx.load_graph(graph)
    ctx.classify(data, want_hits=want_hits)
    if ctx.stats()["non_ascii"]:
        data.tobytes().decode("utf-8")      # UnicodeDecodeError like the reference's text-mode read
    return ctx.counts(), (ctx.hits() if want_hits else None), data


def run(gaf_path, gfa_path, prefix, output_dir=None, device=0):
    """filter-alignments.py main()."""
    out_json, edges_json = output_names(prefix, output_dir)
    graph = Graph.from_files(edges_json, gfa_path)
    ctx = capi.Context(device)
    try:
        counts, recs, data = classify_file(ctx, graph, gaf_path, want_hits=True)
    finally:
        ctx.close()
    capi.write_informative_json(out_json, data, recs, graph.sv_ids)
    return counts, graph
