"""Host side of the alignment filter: the part of filter-alignments.py that is file handling.

    GAF bytes --svjg_classify (HIP)--> per-SV (ref, alt) counts + hit records
              --svjg_write_informative_json (native, host)--> <prefix>_informative_aln.json

The classification itself (filter-alignments.py:123-166) happens in libsvjg_hip.so and the JSON text is produced by
libsvjg_host.so; nothing in this module looks inside an alignment line.
"""
import os
import sys
import threading
import time

import numpy as np

from . import capi, shard
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


CHUNK_BYTES = 1 << 30          # GAF bytes uploaded and classified per call (bounds device memory for the text)
MIN_BYTES_PER_DEVICE = 64 << 20


def read_gaf(path):
    """The file as a read-only uint8 view (memory mapped: nothing is copied on the host; the reference reads it in
    text mode; UTF-8 validity is checked later only if a non-ASCII byte was seen on the device)."""
    if os.path.getsize(path) == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.memmap(path, dtype=np.uint8, mode="r")


def check_utf8(data):
    """UnicodeDecodeError like the reference's text-mode read, without copying the whole file: 64 MB at a time."""
    import codecs
    dec = codecs.getincrementaldecoder("utf-8")()
    n = int(data.size)
    for a in range(0, n, 1 << 26):
        dec.decode(bytes(data[a:a + (1 << 26)]), final=a + (1 << 26) >= n)


def _stamp(t, what):
    """stage timers on stderr when SVJG_VERBOSE is set (measurement only)"""
    if os.environ.get("SVJG_VERBOSE"):
        now = time.perf_counter()
        sys.stderr.write(f"[svjg] {what}: {now - t[0]:.2f} s\n")
        t[0] = now


def pick_devices(n_bytes, device=None):
    """GPUs the alignments are sharded over.  SVJG_DEVICES = "all" | comma-separated indices (an index may repeat: two
    shards on one GPU, used by the tests); unset: every visible GPU that gets at least MIN_BYTES_PER_DEVICE of text."""
    if device is not None:
        return [device]
    spec = os.environ.get("SVJG_DEVICES", "").strip()
    n_vis = max(1, capi.load_library().svjg_device_count())
    if spec and spec != "all":
        return [int(x) for x in spec.split(",") if x.strip() != ""]
    if spec == "all":
        return list(range(n_vis))
    return list(range(max(1, min(n_vis, n_bytes // MIN_BYTES_PER_DEVICE))))


def _classify_shard(dev, graph, data, lo, hi, want_hits, out, r, path=None):
    """One GPU, its contiguous byte range [lo, hi) of the file, chunk by chunk (cuts at line terminators).  With `path`
    the library reads the chunk from the file itself (pinned, double-buffered ingest); `data` (the mapped file) is then
    only looked at around the cut points."""
    try:
        ctx = capi.Context(dev)
        try:
            ctx.load_graph(graph)
            n_chunks = max(1, -(-(hi - lo) // CHUNK_BYTES))
            cuts = [lo + c for c in shard.cut_points(data[lo:hi], n_chunks)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                if b > a and path is not None:
                    ctx.classify_file(path, a, b - a, want_hits=want_hits)
                elif b > a:
                    ctx.classify(data[a:b], base_offset=a, want_hits=want_hits)
            st = ctx.stats()
            out[r] = (ctx.counts().astype(np.uint64), ctx.hits() if want_hits else None, st, None)
        finally:
            ctx.close()
    except BaseException as e:                    # re-raised by the caller: the reference dies at the FIRST bad line of the file
        out[r] = (None, None, None, e)


def classify_file(ctx, graph, gaf_path, want_hits=True):
    """One GPU, one context that keeps the counts (fused driver, tests): -> (counts[n_slots, 2], hit records, the file's bytes)."""
    data = read_gaf(gaf_path)
    ctx.load_graph(graph)
    n = int(data.size)
    cuts = shard.cut_points(data, max(1, -(-n // CHUNK_BYTES)))
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            ctx.classify(data[a:b], base_offset=a, want_hits=want_hits)
    if ctx.stats()["non_ascii"]:
        check_utf8(data)
    return ctx.counts(), (ctx.hits() if want_hits else None), data


def classify_sharded(graph, gaf_path, want_hits=True, devices=None, _t=None):
    """-> (counts[n_slots, 2], hit records, the file's bytes).  The file is cut into one contiguous byte range per GPU
    (line boundaries; rank order = file order), every range is streamed through its GPU in chunks, the per-SV counts are
    summed and the hit records concatenated."""
    t = _t or [time.perf_counter()]
    data = read_gaf(gaf_path)
    devs = devices if devices is not None else pick_devices(int(data.size))
    cuts = shard.cut_points(data, len(devs))
    out = [None] * len(devs)
    if len(devs) == 1:
        _classify_shard(devs[0], graph, data, cuts[0], cuts[1], want_hits, out, 0, gaf_path)
    else:
        th = [threading.Thread(target=_classify_shard, args=(d, graph, data, cuts[r], cuts[r + 1], want_hits, out, r, gaf_path))
              for r, d in enumerate(devs)]
        for x in th:
            x.start()
        for x in th:
            x.join()
    for counts, recs, st, err in out:             # shards are in file order: the first failing shard holds the first bad line
        if err is not None:
            raise err
    _stamp(t, f"tables -> device, upload + classify on {len(devs)} GPU(s)")
    if any(o[2]["non_ascii"] for o in out):
        check_utf8(data)
    total = np.zeros((graph.n_slots, 2), dtype=np.uint64)
    for o in out:
        total += o[0]
    if (total >= 2 ** 32).any():
        raise OverflowError("more than 2^32 informative alignments for one SV")
    recs = None
    if want_hits:
        recs = out[0][1] if len(out) == 1 else np.concatenate([o[1] for o in out])
    _stamp(t, "counts + hit records -> host")
    return total.astype(np.uint32), recs, data


# ---- counts hand-off to predict-genotype.py ---------------------------------------------------------------------------
# predict-genotype.py only needs len() of the two lists of every key (predict-genotype.py:219-226), but the JSON it is
# given is ~5x the GAF.  filter-alignments.py therefore also leaves the key -> (n_ref, n_alt) table in the temp directory,
# tagged with the JSON's path, size and mtime; predict-genotype.py uses it only if all three still match and parses
# the JSON otherwise.  Nothing is written next to the user's files.  SVJG_NO_HANDOFF=1 disables both sides.
def handoff_path(json_path):
    import hashlib
    import tempfile
    h = hashlib.sha1(os.path.abspath(json_path).encode()).hexdigest()[:24]
    return os.path.join(tempfile.gettempdir(), f"svjg_counts_{os.getuid()}_{h}.npz")


def write_handoff(json_path, sv_ids, counts):
    if os.environ.get("SVJG_NO_HANDOFF"):
        return
    try:
        keep = np.flatnonzero(counts.sum(axis=1) > 0)                       # the JSON holds only SVs with an informative alignment
        keys = sorted((sv_ids[i], int(i)) for i in keep)                    # json.dumps(sort_keys=True) order
        st = os.stat(json_path)
        tmp = handoff_path(json_path) + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, keys=np.frombuffer("\0".join(k for k, _ in keys).encode("utf-8"), dtype=np.uint8),
                 counts=counts[[i for _, i in keys]].astype(np.uint32).reshape(-1, 2),
                 tag=np.array([st.st_size, st.st_mtime_ns], dtype=np.int64))
        os.replace(tmp, handoff_path(json_path))
    except OSError:
        pass                                                                 # an optimisation only


def read_handoff(json_path):
    """-> (keys, counts) if a hand-off table for exactly this JSON file exists, else None."""
    if os.environ.get("SVJG_NO_HANDOFF"):
        return None
    try:
        st = os.stat(json_path)
        with np.load(handoff_path(json_path)) as z:
            tag = z["tag"]
            if int(tag[0]) != st.st_size or int(tag[1]) != st.st_mtime_ns:
                return None
            blob = z["keys"].tobytes().decode("utf-8")
            counts = z["counts"].astype(np.uint32).reshape(-1, 2)
        keys = blob.split("\0") if len(counts) else []
        if len(keys) != len(counts):
            return None
        return keys, counts
    except (OSError, ValueError, KeyError):
        return None


def run(gaf_path, gfa_path, prefix, output_dir=None, device=None):
    """filter-alignments.py main()."""
    out_json, edges_json = output_names(prefix, output_dir)
    t = [time.perf_counter()]
    graph = Graph.from_files(edges_json, gfa_path)
    _stamp(t, "edges JSON + GFA -> graph")
    counts, recs, data = classify_sharded(graph, gaf_path, want_hits=True, devices=None if device is None else [device], _t=t)
    capi.write_informative_json(out_json, data, recs, graph.sv_ids)
    _stamp(t, "write _informative_aln.json")
    write_handoff(out_json, graph.sv_ids, counts)
    return counts, graph
