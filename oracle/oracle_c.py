"""ctypes front end of the C oracle (oracle/svjg_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "liboracle.so")
SRC = os.path.join(HERE, "svjg_oracle.c")

HIT_DTYPE = np.dtype([("line_index", "<u8"), ("line_start", "<u8"), ("sv", "<u4"), ("allele", "<u4")])
class Undecided(Exception):
    """a number beyond what the C restatement represents (more than 18 digits): it says so instead of guessing (svjg_oracle.c: ORC_UNDECIDED)"""


_ERR = {1: ValueError, 2: IndexError, 3: KeyError, 4: ZeroDivisionError, 7: Undecided, 9: MemoryError}


def build(force=False):
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-Wall", "-shared", "-fPIC", "-o", SO, SRC], check=True)
    return SO


def _lib():
    lib = ctypes.CDLL(build())
    lib.orc_new.restype = ctypes.c_void_p
    lib.orc_free.argtypes = [ctypes.c_void_p]
    lib.orc_add_edge_entry.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32]
    lib.orc_add_edge_key.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    lib.orc_add_alt_node.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]
    lib.orc_filter.restype = ctypes.c_int
    lib.orc_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                               ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


class COracle:
    """edges: the dict loaded from *_svs_edges.json; alt_len: alt node name -> length."""

    def __init__(self, edges, alt_len):
        self.lib = _lib()
        self.h = self.lib.orc_new()
        self.sv_ids = sorted({e[0] for v in edges.values() for e in v})
        idx = {s: i for i, s in enumerate(self.sv_ids)}
        for key, ents in edges.items():
            kb = key.encode()
            self.lib.orc_add_edge_key(self.h, kb)
            for sv, allele in ents:
                self.lib.orc_add_edge_entry(self.h, kb, idx[sv], int(allele))
        for name, ln in alt_len.items():
            self.lib.orc_add_alt_node(self.h, name.encode(), int(ln))

    def __del__(self):
        try:
            self.lib.orc_free(self.h)
        except Exception:
            pass

    def filter(self, gaf, want_hits=True, hit_cap=None):
        """gaf: bytes / numpy uint8.  -> (counts[n_sv, 2] uint64, hits structured array | None, n_lines)"""
        buf = np.frombuffer(gaf, dtype=np.uint8) if not isinstance(gaf, np.ndarray) else gaf
        counts = np.zeros((len(self.sv_ids), 2), dtype=np.uint64)
        if want_hits:
            cap = hit_cap or max(1024, buf.size // 8)
            hits = np.zeros(cap, dtype=HIT_DTYPE)
            hp = hits.ctypes.data
        else:
            cap, hits, hp = 0, None, None
        nh, nl, el = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self.lib.orc_filter(self.h, buf.ctypes.data if buf.size else None, buf.size, counts.ctypes.data,
                                 len(self.sv_ids), hp, cap, ctypes.byref(nh), ctypes.byref(nl), ctypes.byref(el))
        if rc:
            raise _ERR[rc](f"line {el.value}")
        return counts, (hits[: nh.value] if want_hits else None), nl.value


    def filter_cases(self, frags):
        """Many small GAF fragments (list of bytes), each on its own -> (counts uint64[n, n_sv, 2], exception class or None per fragment)"""
        offs = np.zeros(len(frags) + 1, np.uint64)
        np.cumsum([len(f) for f in frags], out=offs[1:])
        buf = np.frombuffer(b"".join(frags) + b"\0", dtype=np.uint8)
        counts = np.zeros((len(frags), len(self.sv_ids), 2), dtype=np.uint64)
        rc = np.zeros(len(frags), np.int32)
        self.lib.orc_filter_cases.restype = None
        self.lib.orc_filter_cases.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        self.lib.orc_filter_cases(self.h, buf.ctypes.data, offs.ctypes.data, len(frags), counts.ctypes.data, len(self.sv_ids), rc.ctypes.data)
        return counts, [_ERR[int(x)] if x else None for x in rc]


def line_text(gaf_bytes, start):
    """Text the reference stores for the line starting at byte `start`: universal-newline translation,
    then everything before the first 'cg:Z:' (filter-alignments.py:166)."""
    n = len(gaf_bytes)
    e = start
    while e < n and gaf_bytes[e] not in (10, 13):
        e += 1
    s = bytes(gaf_bytes[start:e]).decode("utf-8") + ("\n" if e < n else "")
    return s.split("cg:Z:")[0]


def informative_dict(sv_ids, hits, gaf_bytes):
    out = {}
    cache = {}
    for h in hits:
        st = int(h["line_start"])
        if st not in cache:
            cache[st] = line_text(gaf_bytes, st)
        out.setdefault(sv_ids[int(h["sv"])], [[], []])[int(h["allele"])].append(cache[st])
    return out
