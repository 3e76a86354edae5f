"""Build + call the host harness of the per-line device routines (tests only)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_hostsim.so")
class HostLine(Exception):
    """the exact routine's "the host decides" (svjg.h: SVJG_EXC_ASK_HOST): a decimal column with non-ASCII bytes"""


EXC = {1: ValueError, 2: IndexError, 3: KeyError, 4: ZeroDivisionError, 5: HostLine, 6: TypeError}


def build():
    src = [os.path.join(HERE, "hostsim.cpp"),
           os.path.join(HERE, "..", "..", "svjedi-graph_amd", "csrc", "svjg_line.h"),
           os.path.join(HERE, "..", "..", "svjedi-graph_amd", "csrc", "svjg_host_tables.h"),
           os.path.join(HERE, "..", "..", "svjedi-graph_amd", "csrc", "svjg_planes.h"),
           os.path.join(HERE, "..", "..", "svjedi-graph_amd", "csrc", "svjg_pass.h"),
           os.path.join(HERE, "..", "..", "include", "svjg.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-shared", "-fPIC", "-o", SO, src[0]], check=True)
    return SO


def _lib():
    from svjg import capi
    lib = ctypes.CDLL(build())
    lib.hostsim_classify.restype = ctypes.c_int
    lib.hostsim_classify.argtypes = [ctypes.POINTER(capi.CGraph), ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p] + [ctypes.c_void_p] * 3
    lib.hostsim_check_tables.restype = ctypes.c_uint64
    lib.hostsim_check_tables.argtypes = [ctypes.POINTER(capi.CGraph)]
    return lib, capi


def classify(graph, gaf, tables=True, wave=False):
    """Every line through svjg::slow_line (the exact path the kernels defer to).  tables=False: node names are resolved
    through the sorted node table only (what the device does for names the node-name hash table cannot hold)."""
    lib, capi = _lib()
    cg = capi.cgraph_of(graph)
    if not tables:
        cg.flags |= 2
    if wave:
        # 2 = the two-phase routine of k_classify_slow_wave (64 lanes), 1 = its fallback for very long paths (64 lanes), 3 = one lane with its
        # per-node results kept (k_classify_slow), 4 = the two-phase routine without the table of the pieces' colons (every offset of every piece is tried)
        # 5 = the two-phase routine with every node resolved first and the strands taken by id where the line allows it (k_classify_slow_wave since r05)
        cg.flags |= {1: 4, 2: 8, 3: 32, 4: 8 | 64, 5: 8 | 128}[wave]
    buf = np.frombuffer(gaf, dtype=np.uint8) if not isinstance(gaf, np.ndarray) else gaf
    counts = np.zeros((max(graph.n_slots, 1), 2), dtype=np.uint32)
    nl, eo = ctypes.c_uint64(0), ctypes.c_uint64(0)
    ex = ctypes.c_int(0)
    rc = lib.hostsim_classify(ctypes.byref(cg), buf.ctypes.data if buf.size else None, buf.size, counts.ctypes.data,
                              ctypes.addressof(nl), ctypes.addressof(ex), ctypes.addressof(eo))
    if rc:
        raise EXC[ex.value](f"offset {eo.value}")
    return counts[: graph.n_slots], nl.value


_WAVE_FLAGS = {0: 0, 1: 4, 2: 8, 3: 32, 4: 8 | 64, 5: 8 | 128}


def classify_cases(graph, frags, tables=True, wave=0):
    """Many small GAF fragments (list of bytes), each on its own -> (counts uint32[n, n_slots, 2], exception class or None per fragment)"""
    lib, capi = _lib()
    cg = capi.cgraph_of(graph)
    if not tables:
        cg.flags |= 2
    cg.flags |= _WAVE_FLAGS[int(wave)]
    offs = np.zeros(len(frags) + 1, np.uint64)
    np.cumsum([len(f) for f in frags], out=offs[1:])
    buf = np.frombuffer(b"".join(frags) + b"\0", dtype=np.uint8)
    counts = np.zeros((len(frags), max(graph.n_slots, 1), 2), dtype=np.uint32)
    exc = np.zeros(len(frags), np.int32)
    lib.hostsim_classify_cases.restype = None
    lib.hostsim_classify_cases.argtypes = [ctypes.POINTER(capi.CGraph)] + [ctypes.c_void_p] * 2 + [ctypes.c_uint64] + [ctypes.c_void_p] * 2
    lib.hostsim_classify_cases(ctypes.byref(cg), buf.ctypes.data, offs.ctypes.data, len(frags), counts.ctypes.data, exc.ctypes.data)
    return counts[:, : graph.n_slots], [EXC.get(int(x)) if x else None for x in exc]


def check_tables(graph):
    lib, capi = _lib()
    cg = capi.cgraph_of(graph)
    return lib.hostsim_check_tables(ctypes.byref(cg))


def table_stats(graph):
    """(names left out, names skipped, links left out, name slots, name buckets, link slots, inline hit words, max displacement)"""
    lib, capi = _lib()
    cg = capi.cgraph_of(graph)
    out = (ctypes.c_uint64 * 8)()
    lib.hostsim_table_stats.restype = None
    lib.hostsim_table_stats(ctypes.byref(cg), out)
    return tuple(int(x) for x in out)


def span_classes(text):
    """Byte-class masks of the 64-byte spans of text (len % 64 == 0) by the main kernel's bit-plane routine:
    uint64[n_spans, 8] = nl, cr, tab, ori, nd, dee, colon, high."""
    lib = ctypes.CDLL(build())
    buf = np.frombuffer(bytes(text), dtype=np.uint8)
    assert buf.size % 64 == 0
    out = np.zeros((buf.size // 64, 8), np.uint64)
    lib.hostsim_span_classes.restype = None
    lib.hostsim_span_classes.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    lib.hostsim_span_classes(buf.ctypes.data, buf.size // 64, out.ctypes.data)
    return out


def pass_logic():
    """the fused pass's host decisions (svjedi-graph_amd/csrc/svjg_pass.h) -> (repeat_word(overflow), repeats(has_comm, overflow, guard_sum),
    counts_overflowed(max_ref_sum, max_alt_sum), number of guard words)"""
    lib = ctypes.CDLL(build())
    lib.hostsim_pass_repeat_word.restype = ctypes.c_uint64
    lib.hostsim_pass_repeat_word.argtypes = [ctypes.c_uint32]
    lib.hostsim_pass_repeats.restype = ctypes.c_int
    lib.hostsim_pass_repeats.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64]
    lib.hostsim_pass_counts_overflowed.restype = ctypes.c_int
    lib.hostsim_pass_counts_overflowed.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    lib.hostsim_guard_words.restype = ctypes.c_uint32
    return (lambda o: int(lib.hostsim_pass_repeat_word(o)), lambda c, o, s: bool(lib.hostsim_pass_repeats(int(c), o, s)),
            lambda a, b: bool(lib.hostsim_pass_counts_overflowed(a, b)), int(lib.hostsim_guard_words()))
