"""ctypes binding of libsvjg_hip.so (include/svjg.h).

There is no fallback: if the shared library is missing, or no MI355X is visible, the calls raise.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(HERE), "csrc", "libsvjg_hip.so")
HOST_LIB_PATH = os.path.join(os.path.dirname(HERE), "csrc", "libsvjg_host.so")

HITREC_DT = np.dtype([("line_start", "<u8"), ("slot", "<u4"), ("n_ref", "<u2"), ("n_alt", "<u2")])

EXC_CLASS = {1: ValueError, 2: IndexError, 3: KeyError, 4: ZeroDivisionError, 6: TypeError}
LINE_ERRORS = (ValueError, IndexError, KeyError, ZeroDivisionError, TypeError)      # what a GAF line can make the reference die with
DOVER_TYPE_ERROR = "'>=' not supported between instances of 'int' and 'list'"       # filter-alignments.py:269 under -O (svjg.h: SVJG_EXC_TYPE_ERROR)


class SvjgError(RuntimeError):
    pass


class CGraph(ctypes.Structure):
    _fields_ = [
        ("nodes", ctypes.c_void_p), ("n_nodes", ctypes.c_uint64),
        ("edges", ctypes.c_void_p), ("n_edges", ctypes.c_uint64),
        ("hits", ctypes.c_void_p), ("n_hits", ctypes.c_uint64),
        ("chrom_names", ctypes.c_char_p), ("chrom_off", ctypes.c_void_p), ("chrom_node_lo", ctypes.c_void_p),
        ("n_chrom", ctypes.c_uint32),
        ("n_slots", ctypes.c_uint32), ("d_over", ctypes.c_uint32), ("flags", ctypes.c_uint32),
    ]


class CStats(ctypes.Structure):
    _fields_ = [("n_lines", ctypes.c_uint64), ("n_deferred", ctypes.c_uint64), ("n_hitrecs", ctypes.c_uint64),
                ("non_ascii", ctypes.c_uint64)]


def cgraph_of(g):
    """ctypes view of a svjg.graph.Graph (the Graph must stay alive while the struct is in use)."""
    return CGraph(
        g.nodes.ctypes.data, g.n_nodes, g.edges.ctypes.data, g.n_edges, g.hits.ctypes.data, g.n_hits,
        g.chrom_names, g.chrom_off.ctypes.data, g.chrom_lo.ctypes.data, len(g.chroms),
        g.n_slots, g.d_over, g.flags)


_SIGS = {
    "svjg_abi_version": (ctypes.c_int, []),
    "svjg_device_count": (ctypes.c_int, []),
    "svjg_init": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "svjg_destroy": (None, [ctypes.c_void_p]),
    "svjg_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "svjg_load_graph": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(CGraph)]),
    "svjg_release_host_tables": (None, []),
    "svjg_gaf_upload": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "svjg_gaf_upload_part": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]),
    "svjg_comm_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "svjg_copy_rate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "svjg_classify_resident": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]),
    "svjg_classify": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]),
    "svjg_gaf_upload_file": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64]),
    "svjg_classify_file": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]),
    "svjg_reset_counts": (ctypes.c_int, [ctypes.c_void_p]),
    "svjg_get_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(CStats)]),
    "svjg_get_defer_causes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "svjg_input_error": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]),
    "svjg_get_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]),
    "svjg_set_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]),
    "svjg_alloc_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32]),
    "svjg_get_hits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    "svjg_get_host_lines": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    "svjg_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "svjg_comm_init": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    "svjg_allreduce_counts": (ctypes.c_int, [ctypes.c_void_p]),
    "svjg_comm_init_all": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    "svjg_comm_error": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64]),
    "svjg_allreduce_counts_all": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    "svjg_genotype": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                     ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p]),
    "svjg_genotype_view": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                          ctypes.c_uint32, ctypes.c_double] + [ctypes.POINTER(ctypes.c_void_p)] * 4),
    "svjg_set_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "svjg_run_resident": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_double] + [ctypes.POINTER(ctypes.c_void_p)] * 5),
    "svjg_run_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_double]),
    "svjg_run_end": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_void_p)] * 5),
    "svjg_genotype_boundary": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "svjg_last_kernel_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                           ctypes.POINTER(ctypes.c_float)]),
    "svjg_sync": (ctypes.c_int, [ctypes.c_void_p]),
}

EXPORTS = tuple(_SIGS) + ("svjg_write_informative_json", "svjg_count_informative_json", "svjg_host_free",
                          "svjg_graph_load", "svjg_graph_view", "svjg_graph_info", "svjg_graph_free",
                          "svjg_vcf_load", "svjg_vcf_arrays", "svjg_vcf_write", "svjg_vcf_free")
_lib = None
_host_lib = None


def load_host_library():
    """libsvjg_host.so: the native JSON writer (plain C++, no GPU)."""
    global _host_lib
    if _host_lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise SvjgError(f"{HOST_LIB_PATH} not found: build it first (python __graft_entry__.py build)")
        lib = ctypes.CDLL(HOST_LIB_PATH)
        lib.svjg_write_informative_json.restype = ctypes.c_int
        lib.svjg_write_informative_json.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                                    ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32, ctypes.c_int]
        lib.svjg_count_informative_json.restype = ctypes.c_int
        lib.svjg_count_informative_json.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64),
                                                    ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        lib.svjg_host_free.restype = None
        lib.svjg_host_free.argtypes = [ctypes.c_void_p]
        lib.svjg_graph_load.restype = ctypes.c_int
        lib.svjg_graph_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.svjg_graph_view.restype = ctypes.POINTER(CGraph)
        lib.svjg_graph_view.argtypes = [ctypes.c_void_p]
        lib.svjg_graph_info.restype = ctypes.c_int
        lib.svjg_graph_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
        lib.svjg_graph_free.restype = None
        lib.svjg_graph_free.argtypes = [ctypes.c_void_p]
        lib.svjg_vcf_load.restype = ctypes.c_int
        lib.svjg_vcf_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_void_p)]
        lib.svjg_vcf_arrays.restype = ctypes.c_int
        lib.svjg_vcf_arrays.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_void_p)] * 3 + [ctypes.POINTER(ctypes.c_uint64)]
        lib.svjg_vcf_write.restype = ctypes.c_int
        lib.svjg_vcf_write.argtypes = [ctypes.c_void_p, ctypes.c_char_p] + [ctypes.c_void_p] * 4 + [ctypes.POINTER(ctypes.c_uint64)]
        lib.svjg_vcf_free.restype = None
        lib.svjg_vcf_free.argtypes = [ctypes.c_void_p]
        _host_lib = lib
    return _host_lib


class NativeVcfRows:
    """Rows of a VCF held by libsvjg_host (svjg_vcf_*): the same three arrays as svjg.genotype.VcfRows, and the writer."""

    def __init__(self, lib, h):
        self.lib, self.h = lib, h
        p = [ctypes.c_void_p() for _ in range(3)]
        n = ctypes.c_uint64(0)
        lib.svjg_vcf_arrays(h, ctypes.byref(p[0]), ctypes.byref(p[1]), ctypes.byref(p[2]), ctypes.byref(n))
        n = n.value

        def arr(ptr, dt):
            return np.frombuffer(ctypes.string_at(ptr, n * np.dtype(dt).itemsize), dtype=dt).copy() if n else np.zeros(0, dtype=dt)
        self.sv_type, self.slot, self.ok = arr(p[0], np.uint8), arr(p[1], np.uint32), arr(p[2], np.uint8)

    def write(self, out_path, gt, pl, raw, done):
        gt = np.ascontiguousarray(gt, dtype=np.uint8); pl = np.ascontiguousarray(pl, dtype=np.int64)
        raw = np.ascontiguousarray(raw, dtype=np.uint32); done = np.ascontiguousarray(done, dtype=np.uint8)
        n = len(self.sv_type)
        if len(gt) != n or pl.shape != (n, 3) or raw.shape != (n, 2) or len(done) != n:
            raise ValueError("result arrays do not match the rows")
        nd = ctypes.c_uint64(0)
        rc = self.lib.svjg_vcf_write(self.h, os.fsencode(out_path), gt.ctypes.data, pl.ctypes.data, raw.ctypes.data, done.ctypes.data, ctypes.byref(nd))
        if rc:
            raise OSError(f"cannot write {out_path} ({rc})")
        return nd.value

    def close(self):
        if self.h:
            self.lib.svjg_vcf_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vcf_load_native(vcf_path, keys, slots=None, slot_is_presence=False):
    """NativeVcfRows, or None when the native reader leaves the file to svjg.genotype.VcfRows.  keys: iterable of sv_id strings
    (slots: their count slots, default 0..n-1)."""
    lib = load_host_library()
    keys = list(keys)
    try:
        blob = ("\0".join(keys) + "\0").encode("ascii") if keys else b""
    except UnicodeEncodeError:
        return None
    if blob.count(b"\0") != len(keys):                          # a NUL inside a key
        return None
    sl = None if slots is None else np.ascontiguousarray(slots, dtype=np.uint32)
    h = ctypes.c_void_p()
    rc = lib.svjg_vcf_load(os.fsencode(vcf_path), blob, len(blob), None if sl is None else sl.ctypes.data, len(keys), int(slot_is_presence), ctypes.byref(h))
    if rc == -11:
        return None
    if rc:
        raise SvjgError(f"svjg_vcf_load failed ({rc})")
    return NativeVcfRows(lib, h)


def graph_load_native(edges_json, gfa):
    """Tables of a svjg.graph.Graph from the native loader, or None when it leaves the input to the Python loader.
    -> dict(nodes, edges, hits, n_edges, n_hits, chrom_names (bytes), chrom_off, chrom_lo, sv_ids (list), n_hazard)"""
    lib = load_host_library()
    h = ctypes.c_void_p()
    rc = lib.svjg_graph_load(os.fsencode(edges_json), os.fsencode(gfa), ctypes.byref(h))
    if rc == -11:
        return None
    if rc:
        raise OSError(f"cannot read {edges_json} / {gfa}")
    try:
        v = lib.svjg_graph_view(h).contents
        from .graph import NODE_DT, EDGE_DT

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            return np.frombuffer(ctypes.string_at(ptr, n * np.dtype(dt).itemsize), dtype=dt).copy()
        n_chrom = v.n_chrom
        # (the field is declared c_char_p for the callers that PASS names in; read as such it is a Python bytes object cut at the first
        #  NUL, and string_at() of that would run past its end: take the raw pointer)
        names_ptr = ctypes.c_void_p.from_buffer(v, type(v).chrom_names.offset).value
        chrom_off = arr(v.chrom_off, n_chrom + 1, np.uint32)
        blob, blen, nhz = ctypes.c_void_p(), ctypes.c_uint64(0), ctypes.c_uint32(0)
        lib.svjg_graph_info(h, ctypes.byref(blob), ctypes.byref(blen), ctypes.byref(nhz))
        sv = ctypes.string_at(blob, blen.value).decode("ascii").split("\0")[:-1] if blen.value else []
        return dict(
            nodes=arr(v.nodes, v.n_nodes + 1, NODE_DT), n_nodes=int(v.n_nodes),
            edges=arr(v.edges, max(1, v.n_edges), EDGE_DT), n_edges=int(v.n_edges),
            hits=arr(v.hits, max(1, v.n_hits), np.uint32), n_hits=int(v.n_hits),
            chrom_names=ctypes.string_at(names_ptr, int(chrom_off[-1]) + 4), chrom_off=chrom_off,
            chrom_lo=arr(v.chrom_node_lo, n_chrom + 1, np.uint32), sv_ids=sv, n_hazard=int(nhz.value))
    finally:
        lib.svjg_graph_free(h)


def count_informative_json(path):
    """-> (list of keys in file order, uint32 array [n, 2] of list lengths) of an _informative_aln.json."""
    lib = load_host_library()
    kp, cp = ctypes.c_void_p(), ctypes.c_void_p()
    kl, nk = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = lib.svjg_count_informative_json(os.fsencode(path), ctypes.byref(kp), ctypes.byref(kl), ctypes.byref(cp), ctypes.byref(nk))
    if rc == -10:
        raise ValueError(f"{path}: not the JSON written by filter-alignments.py")
    if rc:
        raise OSError(f"cannot read {path}")
    try:
        blob = ctypes.string_at(kp, kl.value)
        keys = [k.decode("utf-8") for k in blob.split(b"\0")[:-1]] if kl.value else []
        if nk.value:
            cnt = np.ctypeslib.as_array(ctypes.cast(cp, ctypes.POINTER(ctypes.c_uint64)), shape=(nk.value * 2,)).copy()
        else:
            cnt = np.zeros(0, dtype=np.uint64)                      # (nothing to read: the buffer may be empty)
    finally:
        lib.svjg_host_free(kp); lib.svjg_host_free(cp)
    if (cnt >= 2 ** 32).any():
        raise ValueError("list too long")
    return keys, cnt.astype(np.uint32).reshape(-1, 2)


def write_informative_json(path, gaf, recs, sv_ids, n_threads=0):
    """<prefix>_informative_aln.json from hit records (filter-alignments.py:174-175)."""
    lib = load_host_library()
    a = _as_u8(gaf)
    n_threads = n_threads or int(os.environ.get("SVJG_JSON_THREADS", "0"))
    recs = np.ascontiguousarray(recs, dtype=HITREC_DT)
    keys = (ctypes.c_char_p * max(1, len(sv_ids)))(*[s.encode("utf-8") for s in sv_ids])
    rc = lib.svjg_write_informative_json(os.fsencode(path), a.ctypes.data if a.size else None, a.size,
                                         recs.ctypes.data if len(recs) else None, len(recs), keys, len(sv_ids), n_threads)
    if rc == -10:
        raise UnicodeDecodeError("utf-8", b"", 0, 1, "invalid UTF-8 in an informative alignment line")
    if rc:
        raise SvjgError(f"svjg_write_informative_json failed ({rc}) for {path}")


def load_library(path=None):
    """dlopen the HIP library and attach the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SVJG_HIP_LIB") or LIB_PATH    # SVJG_HIP_LIB: a measurement build of the same library (tools/mkvariant.sh -> build/lib_<name>.so)
    if p != LIB_PATH and path is None:
        sys.stderr.write(f"[svjg] measurement build in use: {p}\n")
    if not os.path.exists(p):
        raise SvjgError(f"{p} not found: build it first (python __graft_entry__.py build, needs hipcc); there is no CPU fallback")
    # Copies by shader, not by the copy engines: the ROCm runtime creates the queue of a copy engine the first time a copy happens
    # to be handed to it (6-7 ms each, several engines, at unpredictable calls early in a process: one `svjg_genotype` call of
    # 6.4 ms among 0.13 ms ones).  Must be in the environment before the runtime starts, i.e. before the first HIP call of the
    # process; the price is a slower bulk upload (41 instead of 56 GB/s).  SVJG_SDMA=1 keeps the runtime's default.
    if os.environ.get("SVJG_SDMA", "0") != "1":
        os.environ.setdefault("HSA_ENABLE_SDMA", "0")
    lib = ctypes.CDLL(p)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.svjg_abi_version() != 1:
        raise SvjgError("libsvjg_hip.so ABI mismatch")
    if path is None:
        _lib = lib
    return lib


class Context:
    """One GPU.  Thin, stateful wrapper over the C ABI."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.svjg_init(device, ctypes.byref(h))
        if rc:
            msg = self.lib.svjg_last_error(None)
            raise SvjgError(f"svjg_init failed ({rc}): {msg.decode() if msg else ''}")
        self.h = h
        self.graph = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.svjg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc == -10:
            cls, off = ctypes.c_int(0), ctypes.c_uint64(0)
            self.lib.svjg_input_error(self.h, ctypes.byref(cls), ctypes.byref(off))
            e = TypeError(DOVER_TYPE_ERROR) if cls.value == 6 else EXC_CLASS.get(cls.value, ValueError)(f"malformed GAF line at byte offset {off.value}")
            e.svjg_offset = off.value                           # (svjg/filter.py: reference_error)
            raise e
        if rc == -12:
            raise OverflowError(self.lib.svjg_last_error(self.h).decode())
        if rc:
            raise SvjgError(f"libsvjg_hip error {rc}: {self.lib.svjg_last_error(self.h).decode()}")

    def load_graph(self, g):
        self.graph = g
        self.n_slots = g.n_slots
        cg = cgraph_of(g)
        self._chk(self.lib.svjg_load_graph(self.h, ctypes.byref(cg)))

    def alloc_counts(self, n_slots):
        self.graph = None
        self.n_slots = n_slots
        self._chk(self.lib.svjg_alloc_counts(self.h, n_slots))

    def upload(self, gaf):
        a = _as_u8(gaf)
        self._chk(self.lib.svjg_gaf_upload(self.h, a.ctypes.data if a.size else None, a.size))

    def upload_parts(self, parts, capacity_bytes):
        """the resident text from an iterable of pieces (ascending; at most `capacity_bytes` in all): a text the host never holds
        whole (svjg_gaf_upload_part).  -> bytes uploaded"""
        at, prev = 0, None
        for p in parts:
            if prev is not None:
                self._chk(self.lib.svjg_gaf_upload_part(self.h, prev.ctypes.data if prev.size else None, prev.size, at, capacity_bytes, 0))
                at += int(prev.size)
            prev = _as_u8(p)
        if prev is None:
            prev = np.zeros(0, dtype=np.uint8)
        self._chk(self.lib.svjg_gaf_upload_part(self.h, prev.ctypes.data if prev.size else None, prev.size, at, capacity_bytes, 1))
        return at + int(prev.size)

    def allreduce_on_second_stream(self, on):
        """where the fused pass puts its all-reduce (svjg_comm_set_stream): False = the compute stream (default)"""
        self._chk(self.lib.svjg_comm_set_stream(self.h, 1 if on else 0))

    def copy_rate(self, n_bytes=1 << 31):
        """-> (GB/s read + written by a plain device-to-device copy, GB/s of a kernel that only reads) on this GPU (svjg_copy_rate)"""
        v, r = ctypes.c_double(0), ctypes.c_double(0)
        self._chk(self.lib.svjg_copy_rate(self.h, n_bytes, ctypes.byref(v), ctypes.byref(r)))
        return v.value, r.value

    def classify_resident(self, base_offset=0, want_hits=False):
        self._chk(self.lib.svjg_classify_resident(self.h, base_offset, int(want_hits)))

    def classify(self, gaf, base_offset=0, want_hits=False):
        a = _as_u8(gaf)
        self._chk(self.lib.svjg_classify(self.h, a.ctypes.data if a.size else None, a.size, base_offset, int(want_hits)))

    def classify_file(self, path, offset, n_bytes, want_hits=False):
        """bytes [offset, offset + n_bytes) of the file, read by the library's feeder threads (no host copy here); a read
        error surfaces as OSError like the reference's own open()/read would"""
        rc = self.lib.svjg_classify_file(self.h, os.fsencode(path), offset, n_bytes, int(want_hits))
        if rc == -6:
            raise OSError(self.lib.svjg_last_error(self.h).decode())
        self._chk(rc)

    def reset_counts(self):
        self._chk(self.lib.svjg_reset_counts(self.h))

    def stats(self):
        s = CStats()
        self._chk(self.lib.svjg_get_stats(self.h, ctypes.byref(s)))
        return {"n_lines": s.n_lines, "n_deferred": s.n_deferred, "n_hitrecs": s.n_hitrecs, "non_ascii": s.non_ascii}

    def defer_causes(self):
        """why lines took the exact path (svjg_get_defer_causes)"""
        out = np.zeros(8, dtype=np.uint64)
        self._chk(self.lib.svjg_get_defer_causes(self.h, out.ctypes.data))
        return dict(zip(("columns", "id_tag_filter", "long_path", "node_name", "whole_stripe"), (int(x) for x in out[:5])))

    def counts(self):
        out = np.zeros((self.n_slots, 2), dtype=np.uint32)
        self._chk(self.lib.svjg_get_counts(self.h, out.ctypes.data, self.n_slots))
        return out

    def set_counts(self, c):
        c = np.ascontiguousarray(c, dtype=np.uint32)
        assert c.shape == (self.n_slots, 2)
        self._chk(self.lib.svjg_set_counts(self.h, c.ctypes.data, self.n_slots))

    def hits(self, out=None):
        """the hit records of this context -> a new array, or into `out` (a contiguous slice of HITREC_DT sized stats()["n_hitrecs"]:
        several contexts fill one array side by side, no concatenation afterwards)"""
        n = self.stats()["n_hitrecs"]
        if out is None:
            out = np.empty(n, dtype=HITREC_DT)
        assert out.dtype == HITREC_DT and out.flags["C_CONTIGUOUS"] and len(out) == n
        got = ctypes.c_uint64(0)
        self._chk(self.lib.svjg_get_hits(self.h, out.ctypes.data, n, ctypes.byref(got)))
        assert got.value == n
        return out

    def host_lines(self):
        """byte offsets of the lines the kernels set aside for the host (svjg.h: SVJG_EXC_ASK_HOST), in file order"""
        n = ctypes.c_uint64(0)
        self._chk(self.lib.svjg_get_host_lines(self.h, None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            self._chk(self.lib.svjg_get_host_lines(self.h, out.ctypes.data, n.value, ctypes.byref(n)))
        return np.sort(out)

    def comm_init(self, unique_id, n_ranks, rank):
        self._chk(self.lib.svjg_comm_init(self.h, unique_id, n_ranks, rank))

    def allreduce_counts(self):
        self._chk(self.lib.svjg_allreduce_counts(self.h))

    def genotype(self, sv_type, slot, ok, min_support, err, reuse_outputs=False):
        """-> (gt, pl[n, 3], raw[n, 2], genotyped).  reuse_outputs: the arrays are read-only views of the library's pinned
        result block (svjg_genotype_view: no copy); they are overwritten by the next call and die with the context, so copy
        what has to outlive either."""
        n = len(sv_type)
        sv_type = np.ascontiguousarray(sv_type, dtype=np.uint8)
        slot = np.ascontiguousarray(slot, dtype=np.uint32)
        ok = np.ascontiguousarray(ok, dtype=np.uint8)
        if reuse_outputs and n:
            p = [ctypes.c_void_p() for _ in range(4)]
            self._chk(self.lib.svjg_genotype_view(self.h, sv_type.ctypes.data, slot.ctypes.data, ok.ctypes.data, n, min_support,
                                                  float(err), *[ctypes.byref(x) for x in p]))

            def view(ptr, count, dt):
                a = np.frombuffer((ctypes.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr.value), dtype=dt)
                a.flags.writeable = False
                return a
            return view(p[0], n, np.uint8), view(p[1], n * 3, np.int64).reshape(n, 3), view(p[2], n * 2, np.uint32).reshape(n, 2), view(p[3], n, np.uint8)
        gt = np.empty(n, dtype=np.uint8)
        pl = np.empty((n, 3), dtype=np.int64)
        raw = np.empty((n, 2), dtype=np.uint32)
        done = np.empty(n, dtype=np.uint8)
        self._chk(self.lib.svjg_genotype(self.h, sv_type.ctypes.data, slot.ctypes.data, ok.ctypes.data, n, min_support,
                                         float(err), gt.ctypes.data, pl.ctypes.data, raw.ctypes.data, done.ctypes.data))
        return gt, pl, raw, done

    def set_rows(self, sv_type, slot, ok):
        """the VCF rows' input arrays of genotype(), left on the device for run_resident()"""
        sv_type = np.ascontiguousarray(sv_type, dtype=np.uint8)
        slot = np.ascontiguousarray(slot, dtype=np.uint32)
        ok = np.ascontiguousarray(ok, dtype=np.uint8)
        self._n_rows = len(sv_type)
        self._chk(self.lib.svjg_set_rows(self.h, sv_type.ctypes.data, slot.ctypes.data, ok.ctypes.data, self._n_rows))

    def _run_views(self, p):
        n = self._n_rows
        if not n:
            self.last_boundary = np.zeros(0, np.uint8)
            return np.zeros(0, np.uint8), np.zeros((0, 3), np.int32), np.zeros((0, 2), np.uint32), np.zeros(0, np.uint8)

        def view(ptr, count, dt):
            a = np.frombuffer((ctypes.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr.value), dtype=dt)
            a.flags.writeable = False
            return a
        self.last_boundary = view(p[4], n, np.uint8)
        return view(p[0], n, np.uint8), view(p[1], n * 3, np.int32).reshape(n, 3), view(p[2], n * 2, np.uint32).reshape(n, 2), view(p[3], n, np.uint8)

    def run_resident(self, min_support, err, base_offset=0):
        """One whole pass with one host wait (svjg_run_resident): zero the counts, classify the uploaded text, all-reduce the
        counts if this context has a communicator, genotype the rows of set_rows().  -> (gt, pl[n, 3] int32, raw[n, 2], flags):
        read-only views of the library's pinned result block, overwritten by the second pass after this one; flags bit 0 =
        genotyped, bit 1 = the row's PLs need 64 bits (genotype() has them).  self.last_boundary: the rows to recompute like the
        reference (svjg_genotype_boundary)."""
        p = [ctypes.c_void_p() for _ in range(5)]
        self._chk(self.lib.svjg_run_resident(self.h, base_offset, min_support, float(err), *[ctypes.byref(x) for x in p]))
        return self._run_views(p)

    def run_begin(self, min_support, err, base_offset=0):
        """enqueue a pass (svjg_run_begin); at most two may be in flight"""
        self._chk(self.lib.svjg_run_begin(self.h, base_offset, min_support, float(err)))

    def run_end(self):
        """wait for the oldest pass in flight and return its results like run_resident() (svjg_run_end)"""
        p = [ctypes.c_void_p() for _ in range(5)]
        self._chk(self.lib.svjg_run_end(self.h, *[ctypes.byref(x) for x in p]))
        return self._run_views(p)

    def boundary_flags(self, n_rows):
        """rows of the last genotype() call whose PLs lie within 1e-6 of an integer boundary (to be recomputed by
        svjg.genotype.exact_pl, the reference's own arithmetic)"""
        out = np.zeros(n_rows, dtype=np.uint8)
        if n_rows:
            self._chk(self.lib.svjg_genotype_boundary(self.h, out.ctypes.data, n_rows))
        return out

    def kernel_ms(self):
        a, b, c = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0)
        self._chk(self.lib.svjg_last_kernel_ms(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def sync(self):
        self._chk(self.lib.svjg_sync(self.h))


def release_host_tables():
    """drop the host copy of the kernels' lookup tables that load_graph keeps for the next context with the same graph"""
    load_library().svjg_release_host_tables()


def device_count():
    return int(load_library().svjg_device_count())


def _handles(ctxs):
    return (ctypes.c_void_p * len(ctxs))(*[c.h for c in ctxs])


def comm_init_all(ctxs):
    """One process, one Context per GPU: RCCL communicators for all of them (ncclCommInitAll inside the library)."""
    rc = ctxs[0].lib.svjg_comm_init_all(_handles(ctxs), len(ctxs))
    if rc:                                                       # (its own message buffer: the call may run beside others on these contexts)
        buf = ctypes.create_string_buffer(512)
        ctxs[0].lib.svjg_comm_error(buf, 512)
        raise SvjgError(f"libsvjg_hip error {rc}: {buf.value.decode(errors='replace')}")


def allreduce_counts_all(ctxs):
    """The path's one collective for a single-process run: afterwards every context holds the summed count vector.
    With one context only the overflow guard runs.  OverflowError if a per-SV count does not fit 32 bits."""
    ctxs[0]._chk(ctxs[0].lib.svjg_allreduce_counts_all(_handles(ctxs), len(ctxs)))


def unique_id():
    lib = load_library()
    buf = ctypes.create_string_buffer(128)
    rc = lib.svjg_comm_unique_id(buf)
    if rc:
        raise SvjgError(f"svjg_comm_unique_id failed ({rc})")
    return buf.raw


def _as_u8(x):
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x.view(np.uint8))
    return np.frombuffer(x, dtype=np.uint8)
