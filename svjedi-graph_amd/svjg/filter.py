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
# A further GPU must save more than its communicators cost.  One GPU takes a file in at ~5.7 GB/s end to end (upload + classify beside the
# parts that do not shard), and creating the communicators (ncclCommInitAll) takes seconds — 5.6 s measured for ONE rank on the one-GPU box
# (tools/rccl_probe.py; profiles/r06/rccl_probe.txt); nobody has measured eight.  So the threshold is not a constant but
#     min_bytes_per_device() = rccl_init_s() x INGEST_BYTES_PER_S
# with rccl_init_s() = SVJG_RCCL_INIT_S if set, else what the first multi-GPU run of this user measured and left in
# ~/.cache/svjg/rccl_init_s (classify_sharded writes it), else 5.6: 32 GB per GPU.  BASELINE configs[3] (21.6 GB) therefore takes ONE GPU of
# an eight-GPU node by default — measured sensible on one GPU (the JSON writer bounds that run at 20 of 28 s), unknown on eight.
# SVJG_DEVICES=all | 0,1,... overrides.  (r04: 64 MB — on an eight-GPU node the 2.1 GB of configs[2] would have been cut eight ways.)
INGEST_BYTES_PER_S = 5.7e9
RCCL_INIT_S_DEFAULT = 5.6


def _rccl_init_file():
    return os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "svjg", "rccl_init_s")


def rccl_init_s():
    """seconds ncclCommInitAll takes on this machine: the environment's word, else the last measurement, else the one-rank figure"""
    for src in (lambda: os.environ["SVJG_RCCL_INIT_S"], lambda: open(_rccl_init_file()).read()):
        try:
            v = float(src().split()[0])
            if 0.0 <= v < 3600.0:
                return v
        except (KeyError, OSError, ValueError, IndexError):
            pass
    return RCCL_INIT_S_DEFAULT


def note_rccl_init_s(seconds, n_ranks):
    """what a multi-GPU run measured (best effort: a read-only home changes nothing)"""
    try:
        os.makedirs(os.path.dirname(_rccl_init_file()), exist_ok=True)
        with open(_rccl_init_file(), "w") as fh:
            fh.write(f"{seconds:.3f} {n_ranks} ranks, ncclCommInitAll\n")
    except OSError:
        pass


def min_bytes_per_device():
    return max(64 << 20, int(rccl_init_s() * INGEST_BYTES_PER_S))


MIN_BYTES_PER_DEVICE = int(RCCL_INIT_S_DEFAULT * INGEST_BYTES_PER_S)          # (the default's value, 32 GB: documentation and tests)


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


def first_utf8_error(data, limit=None):
    """Byte offset of the first invalid UTF-8 sequence in data[:limit], or None."""
    import codecs
    dec = codecs.getincrementaldecoder("utf-8")()
    n = int(data.size) if limit is None else min(int(data.size), int(limit))
    step = 1 << 26
    for a in range(0, n, step):
        try:
            dec.decode(bytes(data[a:min(a + step, n)]), final=a + step >= n and n == int(data.size))
        except UnicodeDecodeError as e:
            # e.start is relative to what the decoder was given now plus what it had kept back (< 4 bytes)
            return max(0, a + e.start - 3) if e.start < 3 and a else a + e.start
    return None


def reference_error(data, exc):
    """Which exception the reference dies with when a line is malformed (`exc`, raised by the library with the line's offset)
    AND the file holds bytes that are not UTF-8: the reference reads the GAF in text mode, in blocks of 8192 bytes that are
    decoded as they are read, so the UnicodeDecodeError of a block comes before the ValueError / IndexError / ... of any line
    that ends in or behind that block (filter-alignments.py:123-126)."""
    off = getattr(exc, "svjg_offset", None)
    if off is None:
        return exc
    n = int(data.size)
    e = int(off)
    while e < n and data[e] not in (10, 13):                # the offending line's terminator
        blk = np.asarray(data[e:e + 65536])
        hit = np.flatnonzero((blk == 10) | (blk == 13))
        if hit.size:
            e += int(hit[0])
            break
        e += blk.size
    bad = first_utf8_error(data, min(n, (e // 8192 + 1) * 8192 + 4))
    if bad is not None and e >= (bad // 8192) * 8192:
        try:
            bytes(data[bad:bad + 4]).decode("utf-8")
        except UnicodeDecodeError as u:
            return u
    return exc


# ---- lines the kernels leave to the host (svjg.h: SVJG_EXC_ASK_HOST) ---------------------------------------------------------
# Two things the reference's int() / float() / str.rstrip() (filter-alignments.py:125, :189-194) do that the kernels do not:
#   * they take Unicode digits and blanks: a line whose decimal column (or id:f: value) fails the ASCII rules and holds a byte >= 0x80;
#   * Python's integers have no width (and CPython >= 3.10.7 refuses a literal of more than sys.get_int_max_str_digits() = 4300 digits
#     with a ValueError): a decimal column of more than 18 digits (r06; the kernels' integers are 64 bits wide).
# Such a line is set aside by the kernels: not counted, not an error.  Here it is put to Python's own int() / float() — the interpreter
# the reference would run in —: it either raises what the reference raises, or is rewritten into an ASCII spelling within the kernels'
# range that means the same to the reference and goes through the GPU again (its hit records are mapped back onto the original line,
# whose text the JSON holds).
HOST_BASE = 1 << 60            # base offset of the resubmitted lines: hit records / errors at or beyond it belong to them
_INT_COLS = (1, 2, 3, 6, 7, 8, 9, 10, 11)
_COL_CAP = 10 ** 18 - 1        # svjg_line.h: COL_DIGITS = 18


class UnsupportedLine(RuntimeError):
    """A GAF line the reference reads and this implementation refuses (exit 1 instead of a guess): a path node that is no node
    of the graph whose name holds a coordinate of more than 12 digits or non-ASCII digits, or a path of more than 65 536 nodes."""


# what resolve_host_lines can raise for a line: the reference's exceptions (OverflowError: Am / Alen beyond a double, :196) or the refusal
HOST_LINE_ERRORS = capi.LINE_ERRORS + (OverflowError, UnsupportedLine)


def _ascii_number(s):
    import unicodedata
    out = []
    for ch in s:
        if ord(ch) < 128:
            out.append(ch)
        elif ch.isspace():
            out.append(" ")
        else:
            out.append(str(unicodedata.decimal(ch)))           # (int() / float() accepted the text: what is left are decimal digits)
    return "".join(out)


def _clamp(v):
    return max(-_COL_CAP, min(_COL_CAP, v))


def host_line(text):
    """read_gaf_line (filter-alignments.py:184-198) on one line as the text-mode read delivers it, with Python's own int() /
    float().  Raises what the reference raises there; else -> the ASCII spelling of the line for the kernels."""
    line = text.rstrip()
    cols = line.split("\t")
    Qid, Qlen, Qs, Qe = cols[:4]                                # (ValueError: not enough values to unpack, like the reference)
    Tid, Tlen, Ts, Te = cols[5:9]
    Am, Alen, Aq = cols[9:12]
    val = {i: int(cols[i]) for i in _INT_COLS}
    if "id:f:" in line:
        float(line.split("id:f:")[-1].split("\t")[0])
    else:
        val[9] / val[10]                                        # (ZeroDivisionError; OverflowError for a quotient beyond a double)
    for i in _INT_COLS:
        cols[i] = _ascii_number(cols[i])
        if sum(ch.isdigit() for ch in cols[i]) > 18 and abs(val[i]) <= _COL_CAP:
            cols[i] = str(val[i])                               # (more than 18 digits for a value that needs fewer: "0000000000000000000005", "1_0_0_...")
    if any(abs(v) > _COL_CAP for v in val.values()):
        # Values beyond the kernels' 18 digits.  Six of the nine columns only have to BE integers (:185-191) and Alen only to be non-zero
        # without an id:f: tag (:196); Tlen, Ts, Te enter two comparisons (:260-271):
        #     left_sum - Ts >= d_over    and    right_sum - (Tlen - Te - 1) >= d_over
        # with sums of node lengths that the kernels keep below 10^17 in magnitude (svjg_line.h: NAME_DIGITS).  Clamping Ts and
        # U = Tlen - Te - 1 to +-(10^18 - 1) leaves both outcomes as they are; U is formed here, exactly, and written as Tlen = U + 1, Te = 0.
        for i in (1, 2, 3, 9, 11):
            if abs(val[i]) > _COL_CAP:
                cols[i] = "1"
        if abs(val[10]) > _COL_CAP:
            cols[10] = "1"
        if max(abs(val[6]), abs(val[7]), abs(val[8])) > _COL_CAP:
            u = _clamp(val[6] - val[8] - 1)
            cols[6], cols[7], cols[8] = (str(u + 1), str(_clamp(val[7])), "0") if u < _COL_CAP else (str(u), str(_clamp(val[7])), "-1")
    out = "\t".join(cols)
    if "id:f:" in out:
        head, _, tail = out.rpartition("id:f:")
        v, sep, rest = tail.partition("\t")
        if not v.isascii():
            if head.count("\t") == 5:
                # The tag's value is a piece of the PATH column: rewriting it would rename a node.  float() has accepted it (above), and
                # the reference looks at the LAST "id:f:" of the line only (:194): one more tag with an ASCII value behind the line's
                # columns leaves every column as it is and gives the kernels a value they read.
                out += "\tid:f:1"
            else:
                out = head + "id:f:" + _ascii_number(v) + sep + rest
    return out.encode("utf-8") + b"\n"


def _name_error(line_bytes):
    """A line the exact routine set aside TWICE: every column is plain, so it met a node name it could not read in a sum the reference forms —
    the name of no graph node (those are canonical ASCII) with a coordinate of more than 12 digits or with bytes >= 0x80.  If Python's own
    arithmetic on EVERY such name of the path (get_node_len, filter-alignments.py:343-349: int(end) first, then int(start)) dies with one and the
    same exception class, the reference dies with it whichever of them it met: that exception.  Else (some such name is a number Python computes
    with) None: the caller refuses the line."""
    import re
    try:
        path = line_bytes.decode("utf-8").split("\t")[5]
    except (UnicodeDecodeError, IndexError):
        return None
    names = [n for n in re.split("[<>]", path) if n] if path[:1] in "<>" else [n[:-1] for n in path.split(",") if n]
    classes = set()
    for nm in names:
        coords = nm.split(":")[-1]
        if "." in coords:
            continue                                            # (an insertion node: its length is the GFA's, not arithmetic)
        if coords.isascii() and all(len(run) <= 12 for run in re.findall("[0-9_]+", coords)):
            continue
        try:
            parts = coords.split("-")
            int(parts[1]) - int(parts[0]) + 1
            return None                                         # a number the reference computes with and the kernels cannot hold
        except (ValueError, IndexError) as e:
            classes.add(type(e))
    if len(classes) == 1:
        return classes.pop()("node name the reference cannot read either (filter-alignments.py:343-349)")
    return None


def resolve_host_lines(ctxs, data, want_hits, error=None):
    """After every shard is classified (`error`: the exception of the first bad line the kernels met, if any): decide the lines the
    kernels set aside.  Raises the exception of the file's first bad line (a host line's own, or `error`); else resubmits the
    accepted lines through ctxs[0] and returns (their offsets in the resubmitted text, the original offsets) for remap_hits(),
    or None when there was nothing to do."""
    offs = np.sort(np.concatenate([c.host_lines() for c in ctxs])) if ctxs else np.zeros(0, np.uint64)
    limit = None if error is None else getattr(error, "svjg_offset", None)
    first = error
    accepted = []
    n = int(data.size)
    for off in offs.tolist():
        if limit is not None and off >= limit:
            break
        e = off
        while e < n and data[e] not in (10, 13):
            blk = np.asarray(data[e:e + 65536])
            hit = np.flatnonzero((blk == 10) | (blk == 13))
            if hit.size:
                e += int(hit[0])
                break
            e += blk.size
        raw = bytes(data[off:e])
        try:
            text = raw.decode("utf-8") + ("\n" if e < n else "")
        except UnicodeDecodeError:
            break                                               # (the text-mode read dies first: check_utf8 / reference_error say where)
        try:
            accepted.append((off, host_line(text)))
        except (ValueError, ZeroDivisionError, IndexError, OverflowError) as ex:
            ex.svjg_offset = off
            first, limit = ex, off
            break
    if accepted:
        starts = np.cumsum([0] + [len(b) for _, b in accepted[:-1]]).astype(np.uint64)
        orig = np.array([o for o, _ in accepted], dtype=np.uint64)
        ctx = ctxs[0]
        before = len(ctx.host_lines())
        try:
            ctx.classify(np.frombuffer(b"".join(b for _, b in accepted), dtype=np.uint8), base_offset=HOST_BASE, want_hits=want_hits)
        except capi.LINE_ERRORS as ex:
            o = getattr(ex, "svjg_offset", None)
            if o is not None and o >= HOST_BASE:
                ex.svjg_offset = int(orig[np.searchsorted(starts, np.uint64(o - HOST_BASE), side="right") - 1])
                if limit is None or ex.svjg_offset < limit:
                    first = ex
        again = ctx.host_lines()[before:]
        if len(again):
            # every column of a rewritten line is plain ASCII within range: what the exact routine could not read is a NODE NAME — a path
            # node that is no node of the graph (its length is arithmetic on its name, filter-alignments.py:343-349) with a coordinate of
            # more than 12 digits or of non-ASCII digits, in a sum the reference forms.  The reference computes on; this implementation
            # refuses unless an earlier line is fatal anyway.
            k = int(np.searchsorted(starts, np.uint64(int(np.min(again)) - HOST_BASE), side="right") - 1)
            o = int(orig[k])
            if limit is None or o < limit:
                ex = _name_error(accepted[k][1]) or UnsupportedLine(f"GAF line at byte offset {o}: a path node that is no node of the graph has a coordinate of more than 12 digits "
                                     "(or non-ASCII digits), or the path has more than 65 536 nodes; the reference computes with it, this implementation "
                                     "does not (DESIGN.md section 8)")
                ex.svjg_offset = o
                if first is None or getattr(first, "svjg_offset", None) is None or o < first.svjg_offset:
                    first = ex
    if first is not None:
        raise first
    return (starts, orig) if accepted else None


def remap_hits(recs, remap):
    """hit records of resubmitted host lines -> the original lines' offsets"""
    if remap is None or recs is None or len(recs) == 0:
        return recs
    starts, orig = remap
    sel = recs["line_start"] >= HOST_BASE
    if sel.any():
        recs = recs.copy()
        idx = np.searchsorted(starts, recs["line_start"][sel] - np.uint64(HOST_BASE), side="right") - 1
        recs["line_start"][sel] = orig[idx]
    return recs



def _stamp(t, what):
    """stage timers on stderr when SVJG_VERBOSE is set (measurement only)"""
    if os.environ.get("SVJG_VERBOSE"):
        now = time.perf_counter()
        sys.stderr.write(f"[svjg] {what}: {now - t[0]:.2f} s\n")
        t[0] = now


def pick_devices(n_bytes, device=None):
    """GPUs the alignments are sharded over.  SVJG_DEVICES = "all" | comma-separated indices (an index may repeat: two
    shards on one GPU, used by the tests); unset: as many of the visible GPUs as get at least min_bytes_per_device() of text each (with
    the default communicator time: one GPU for anything below 64 GB — see there)."""
    if device is not None:
        return [device]
    spec = os.environ.get("SVJG_DEVICES", "").strip()
    n_vis = max(1, capi.load_library().svjg_device_count())
    if spec and spec != "all":
        return [int(x) for x in spec.split(",") if x.strip() != ""]
    if spec == "all":
        return list(range(n_vis))
    return list(range(max(1, min(n_vis, n_bytes // min_bytes_per_device()))))


def _classify_on_device(ctx, graph, data, ranges, want_hits, out, r, path=None):
    """One GPU (one Context that keeps adding to its count vector), its contiguous byte ranges [lo, hi) of the file in file
    order, each streamed chunk by chunk (cuts at line terminators).  With `path` the library reads the chunk from the file
    itself (pinned, double-buffered ingest); `data` (the mapped file) is then only looked at around the cut points."""
    try:
        ctx.load_graph(graph)
        for lo, hi in ranges:
            n_chunks = max(1, -(-(hi - lo) // CHUNK_BYTES))
            cuts = [lo + c for c in shard.cut_points(data[lo:hi], n_chunks)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                if b > a and path is not None:
                    ctx.classify_file(path, a, b - a, want_hits=want_hits)
                elif b > a:
                    ctx.classify(data[a:b], base_offset=a, want_hits=want_hits)
        out[r] = None
    except BaseException as e:                    # re-raised by the caller: the reference dies at the FIRST bad line of the file
        out[r] = e


class _CommInit(threading.Thread):
    """RCCL communicators for the contexts of this process (ncclCommInitAll), created in a thread of its own WHILE the GPUs upload and
    classify: over eight ranks that call takes seconds — the order of the whole run at BASELINE configs[2] — and nothing it does needs
    the contexts to be idle (it sets ctx->comm, which only the all-reduce behind the classification reads).  join_or_raise() before that
    all-reduce; `seconds` = how long the call took."""

    def __init__(self, ctxs):
        super().__init__(daemon=True)
        self.ctxs, self.error, self.seconds = ctxs, None, None

    def run(self):
        t = time.perf_counter()
        try:
            capi.comm_init_all(self.ctxs)
        except BaseException as e:                # noqa: BLE001 (re-raised by join_or_raise)
            self.error = e
        self.seconds = time.perf_counter() - t

    def join_or_raise(self, started=True):
        if started:
            self.join()
        if self.error is not None:
            raise self.error
        return self.seconds


def classify_file(ctx, graph, gaf_path, want_hits=True):
    """One GPU, one context that keeps the counts (fused driver, tests): -> (counts[n_slots, 2], hit records, the file's bytes)."""
    data = read_gaf(gaf_path)
    ctx.load_graph(graph)
    n = int(data.size)
    cuts = shard.cut_points(data, max(1, -(-n // CHUNK_BYTES)))
    err = None
    try:
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                ctx.classify(data[a:b], base_offset=a, want_hits=want_hits)
    except capi.LINE_ERRORS as e:
        err = e
    try:
        remap = resolve_host_lines([ctx], data, want_hits, err)
    except HOST_LINE_ERRORS as e:
        raise reference_error(data, e)
    if ctx.stats()["non_ascii"]:
        check_utf8(data)
    capi.allreduce_counts_all([ctx])              # one GPU: only the overflow guard of the 32-bit count fields
    return ctx.counts(), (remap_hits(ctx.hits(), remap) if want_hits else None), data


def classify_sharded(graph, gaf_path, want_hits=True, devices=None, _t=None):
    """-> (counts[n_slots, 2], hit records, the file's bytes).  The file is cut into one contiguous byte range per entry of
    `devices` (line boundaries; shard order = file order); every GPU streams its ranges through ONE context, so its count
    vector holds the sum of its shards, and the per-GPU vectors are summed by the path's one collective, an RCCL all-reduce
    inside the library (svjg_allreduce_counts_all: ncclCommInitAll over the contexts of this process).  Hit records stay per
    GPU and are concatenated (the JSON writer orders them by line offset)."""
    t = _t or [time.perf_counter()]
    data = read_gaf(gaf_path)
    devs = devices if devices is not None else pick_devices(int(data.size))
    cuts = shard.cut_points(data, len(devs))
    distinct = list(dict.fromkeys(devs))
    ranges = {d: [(cuts[r], cuts[r + 1]) for r, x in enumerate(devs) if x == d] for d in distinct}
    ctxs = []
    comm = None
    try:
        for d in distinct:
            ctxs.append(capi.Context(d))
        if len(ctxs) > 1:
            # The communicators: by default created HERE, before any context uploads or classifies (seconds on the critical path of a
            # multi-GPU run, which pick_devices only chooses for >= 32 GB a GPU).  SVJG_COMM_OVERLAP=1 creates them in a thread of its own
            # beside upload + classify instead — measured with stand-in contexts only (tests/test_comm_overlap.py): ncclCommInitAll beside
            # hipMalloc / hipFree / kernel launches on the same devices has never run on hardware, so it stays opt-in until it has.
            comm = _CommInit(ctxs)
            if os.environ.get("SVJG_COMM_OVERLAP"):
                comm.start()
            else:
                comm.run()
                t_init = comm.join_or_raise(started=False)
                note_rccl_init_s(t_init, len(ctxs))
                if os.environ.get("SVJG_VERBOSE"):
                    sys.stderr.write(f"[svjg] RCCL communicators for {len(ctxs)} GPUs (ncclCommInitAll): {t_init:.2f} s, in front of upload + classify\n")
                comm = None
        errs = [None] * len(distinct)
        if len(distinct) == 1:
            _classify_on_device(ctxs[0], graph, data, ranges[distinct[0]], want_hits, errs, 0, gaf_path)
        else:
            th = [threading.Thread(target=_classify_on_device, args=(ctxs[i], graph, data, ranges[d], want_hits, errs, i, gaf_path))
                  for i, d in enumerate(distinct)]
            for x in th:
                x.start()
            for x in th:
                x.join()
        # the first failing shard (file order) holds the first bad line
        first_bad = [(ranges[d][0][0], errs[i]) for i, d in enumerate(distinct) if errs[i] is not None]
        err = min(first_bad, key=lambda x: x[0])[1] if first_bad else None
        if err is not None and not isinstance(err, capi.LINE_ERRORS):
            raise err
        try:
            remap = resolve_host_lines(ctxs, data, want_hits, err)       # (lines with non-ASCII digits: Python's int() decides)
        except HOST_LINE_ERRORS as e:
            raise reference_error(data, e)
        capi.release_host_tables()                 # (every context has the graph: the shared host copy of the kernels' tables can go)
        _stamp(t, f"tables -> device, upload + classify on {len(distinct)} GPU(s)")
        if any(c.stats()["non_ascii"] for c in ctxs):
            check_utf8(data)
        if comm is not None:
            t_wait = time.perf_counter()
            t_init = comm.join_or_raise()
            note_rccl_init_s(t_init, len(ctxs))
            if os.environ.get("SVJG_VERBOSE"):
                sys.stderr.write(f"[svjg] RCCL communicators for {len(ctxs)} GPUs (ncclCommInitAll): {t_init:.2f} s beside upload + classify, "
                                 f"{time.perf_counter() - t_wait:.2f} s of it waited for here\n")
            comm = None
        capi.allreduce_counts_all(ctxs)           # (one GPU: only the overflow guard)
        total = ctxs[0].counts()
        recs = None
        if want_hits:
            recs = remap_hits(gather_hits(ctxs), remap)
        _stamp(t, "count all-reduce, counts + hit records -> host")
        return total, recs, data
    finally:
        if comm is not None and comm.is_alive():   # (an error on the way: the communicator call must have returned before its contexts go)
            comm.join()
        for c in ctxs:
            c.close()


# ---- counts hand-off to predict-genotype.py ---------------------------------------------------------------------------
# predict-genotype.py only needs len() of the two lists of every key (predict-genotype.py:219-226), but the JSON it is
# given is ~5x the GAF.  filter-alignments.py therefore also leaves the key -> (n_ref, n_alt) table in a directory only
# this user can write (mode 0700 under $XDG_CACHE_HOME / ~/.cache), tagged with the JSON's size, mtime and a digest of its
# first and last megabyte; predict-genotype.py uses it only if the directory and the file belong to the caller, nobody else
# may write them, and all three tags still match — and parses the JSON otherwise.  Nothing is written next to the user's
# files.  SVJG_NO_HANDOFF=1 disables both sides.
def _handoff_dir(create):
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    d = os.path.join(base, "svjedi-graph_amd")
    if create:
        os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise OSError(f"{d} is not a private directory of this user")
    return d


def handoff_path(json_path, create=False):
    import hashlib
    h = hashlib.sha1(os.path.abspath(json_path).encode()).hexdigest()[:24]
    return os.path.join(_handoff_dir(create), f"counts_{h}.npz")


def _json_digest(json_path, size):
    """sha1 over the first and the last MB of the file and its size: binds the table to the JSON's content, not only to its
    size and time stamp (a same-size rewrite on a file system with coarse mtimes would otherwise pass)."""
    import hashlib
    h = hashlib.sha1(str(size).encode())
    with open(json_path, "rb") as f:
        h.update(f.read(1 << 20))
        if size > (1 << 20):
            f.seek(max(1 << 20, size - (1 << 20)))
            h.update(f.read(1 << 20))
    return np.frombuffer(h.digest(), dtype=np.uint8)


def write_handoff(json_path, sv_ids, counts):
    if os.environ.get("SVJG_NO_HANDOFF"):
        return
    target = None
    try:
        target = handoff_path(json_path, create=True)
        keep = np.flatnonzero(counts.sum(axis=1) > 0)                       # the JSON holds only SVs with an informative alignment
        keys = sorted((sv_ids[i], int(i)) for i in keep)                    # json.dumps(sort_keys=True) order
        st = os.stat(json_path)
        tmp = target + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, keys=np.frombuffer("\0".join(k for k, _ in keys).encode("utf-8"), dtype=np.uint8),
                 counts=counts[[i for _, i in keys]].astype(np.uint32).reshape(-1, 2),
                 tag=np.array([st.st_size, st.st_mtime_ns], dtype=np.int64), digest=_json_digest(json_path, st.st_size))
        os.replace(tmp, target)
    except OSError:
        # an optimisation only — but a stale table must not outlive a failed update
        try:
            if target:
                os.unlink(target)
        except OSError:
            pass


def read_handoff(json_path):
    """-> (keys, counts) if a hand-off table written by this user for exactly this JSON file exists, else None."""
    if os.environ.get("SVJG_NO_HANDOFF"):
        return None
    try:
        st = os.stat(json_path)
        p = handoff_path(json_path)
        fst = os.lstat(p)
        import stat
        if not stat.S_ISREG(fst.st_mode) or fst.st_uid != os.getuid() or (fst.st_mode & 0o022):
            return None
        with np.load(p) as z:
            tag = z["tag"]
            if int(tag[0]) != st.st_size or int(tag[1]) != st.st_mtime_ns:
                return None
            if not np.array_equal(z["digest"], _json_digest(json_path, st.st_size)):
                return None
            blob = z["keys"].tobytes().decode("utf-8")
            counts = z["counts"].astype(np.uint32).reshape(-1, 2)
        keys = blob.split("\0") if len(counts) else []
        if len(keys) != len(counts):
            return None
        return keys, counts
    except (OSError, ValueError, KeyError):
        return None


STREAM_BLOCK = 8 << 20          # bytes asked of a stream at once
STREAM_CHUNK = int(os.environ.get("SVJG_STREAM_CHUNK", 64 << 20))   # whole lines are classified as soon as this much new text has arrived


def gather_hits(ctxs):
    """the hit records of all contexts in one array, context after context: every GPU copies its records into its part of the
    array (one thread per context, the copies run side by side; no concatenation of per-GPU arrays — 5.7 GB at configs[3])"""
    if len(ctxs) == 1:
        return ctxs[0].hits()
    ns = [c.stats()["n_hitrecs"] for c in ctxs]
    at = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    recs = np.empty(int(at[-1]), dtype=capi.HITREC_DT)
    errs = []

    def work(i):
        try:
            ctxs[i].hits(recs[at[i]:at[i + 1]])
        except BaseException as e:                       # noqa: BLE001 (re-raised by the caller's thread)
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    if errs:
        raise errs[0]
    return recs


def classify_stream(graph, stream, want_hits=True, device=0, _t=None):
    """-a - : the GAF comes through a pipe (`minigraph ... | filter-alignments.py -a - ...`, svjedi-graph.py:100-105 appends
    minigraph's output to a file first).  Whole lines are classified while the producer is still writing: every
    STREAM_CHUNK bytes the text up to the last line terminator goes through the GPU; the bytes are kept (the JSON holds the
    line texts) and returned like classify_sharded's mapped file.  -> (counts, hit records, the stream's bytes)"""
    t = _t or [time.perf_counter()]
    buf = bytearray()
    done = 0                                         # bytes of buf already classified (a line boundary)
    ctx = capi.Context(device)
    try:
        ctx.load_graph(graph)

        def flush(upto):
            nonlocal done
            if upto > done:
                view = np.frombuffer(buf, dtype=np.uint8, count=upto - done, offset=done)
                try:
                    ctx.classify(view, base_offset=done, want_hits=want_hits)
                finally:
                    del view                       # (a bytearray with a live export cannot grow)
                done = upto
        err = None
        try:
            while True:
                b = stream.read(STREAM_BLOCK)
                if not b:
                    break
                buf += b
                if len(buf) - done >= STREAM_CHUNK:
                    cut = buf.rfind(b"\n", done) + 1          # (a lone \r ends a line too, but never needs to end a chunk)
                    flush(cut)
            flush(len(buf))
        except capi.LINE_ERRORS as e:
            # the reference would have met this line only after reading everything in front of it; read on so that a
            # non-UTF-8 byte before it still wins (reference_error), then report
            for b in iter(lambda: stream.read(STREAM_BLOCK), b""):
                buf += b
            err = e
        data = np.frombuffer(bytes(buf), dtype=np.uint8) if err is not None else np.frombuffer(buf, dtype=np.uint8)
        try:
            remap = resolve_host_lines([ctx], data, want_hits, err)
        except HOST_LINE_ERRORS as e:
            raise reference_error(data, e)
        _stamp(t, "stream -> device, classified while it arrived")
        if ctx.stats()["non_ascii"]:
            check_utf8(data)
        capi.allreduce_counts_all([ctx])          # one GPU: the overflow guard
        return ctx.counts(), (remap_hits(ctx.hits(), remap) if want_hits else None), data
    finally:
        ctx.close()


def run(gaf_path, gfa_path, prefix, output_dir=None, device=None, dover_given=False):
    """filter-alignments.py main().  dover_given: -O was on the command line — the reference then holds a list where it expects a
    number and dies with TypeError at the first link that has a candidate SV (:153 -> :269); a GAF without one is written as usual."""
    out_json, edges_json = output_names(prefix, output_dir)
    t = [time.perf_counter()]
    graph = Graph.from_files(edges_json, gfa_path)
    if dover_given:
        from .graph import GRAPH_DOVER_LIST
        graph.flags |= GRAPH_DOVER_LIST
    _stamp(t, "edges JSON + GFA -> graph")
    stream = sys.stdin.buffer if gaf_path == "-" else None
    if stream is None:
        with open(gaf_path, "rb") as fh:
            gz = fh.read(2) == b"\x1f\x8b"
        if gz:
            # An extension: a gzip-compressed GAF is inflated on the fly and classified while it is being read, like a
            # pipe.  The reference cannot read such a file (its text-mode open() dies with UnicodeDecodeError on the second byte), so no
            # input it accepts is treated differently.
            import gzip
            stream = gzip.open(gaf_path, "rb")
    if stream is not None:
        try:
            counts, recs, data = classify_stream(graph, stream, want_hits=True, device=device or 0, _t=t)
        finally:
            if stream is not sys.stdin.buffer:
                stream.close()
        capi.write_informative_json(out_json, data, recs, graph.sv_ids)
        _stamp(t, "write _informative_aln.json")
        write_handoff(out_json, graph.sv_ids, counts)
        return counts, graph
    counts, recs, data = classify_sharded(graph, gaf_path, want_hits=True, devices=None if device is None else [device], _t=t)
    capi.write_informative_json(out_json, data, recs, graph.sv_ids)
    _stamp(t, "write _informative_aln.json")
    write_handoff(out_json, graph.sv_ids, counts)
    return counts, graph
