"""The counts hand-off between the two drop-in scripts (svjg/filter.py: write_handoff / read_handoff): it must describe
exactly what a parse of the JSON gives, must be ignored as soon as the JSON file is not the one it was written for, and must
live where only the calling user can write."""
import json
import os
import shutil
import stat

import numpy as np

from svjg import capi, filter as flt


def _case(golden, tmp_path):
    src = f"{golden}/testdir/ref_informative_aln.json"
    dst = str(tmp_path / "p_informative_aln.json")
    shutil.copy(src, dst)
    ref = json.load(open(dst))
    sv_ids = sorted(ref) + ["chrZ:DEL-1-2"]               # one SV without informative alignments: not in the JSON
    counts = np.array([[len(ref[k][0]), len(ref[k][1])] for k in sorted(ref)] + [[0, 0]], dtype=np.uint32)
    return dst, sv_ids, counts


def test_handoff_matches_the_json(golden, tmp_path, monkeypatch):
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    dst, sv_ids, counts = _case(golden, tmp_path)
    perm = np.random.default_rng(1).permutation(len(sv_ids))    # slots are in graph order, not in key order
    flt.write_handoff(dst, [sv_ids[i] for i in perm], counts[perm])
    d = tmp_path / "cache" / "svjedi-graph_amd"
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700            # a directory of the user's own, nothing in the shared temp dir
    got = flt.read_handoff(dst)
    assert got is not None
    keys, cnt = got
    k2, c2 = capi.count_informative_json(dst)
    assert keys == k2 and np.array_equal(cnt, c2)
    # the file changes: the table no longer applies
    with open(dst, "a") as fh:
        fh.write(" ")
    assert flt.read_handoff(dst) is None
    flt.write_handoff(dst, sv_ids, counts)
    os.utime(dst, ns=(1, 1))
    assert flt.read_handoff(dst) is None
    monkeypatch.setenv("SVJG_NO_HANDOFF", "1")
    flt.write_handoff(dst, sv_ids, counts)
    assert flt.read_handoff(dst) is None


def test_handoff_is_bound_to_content_and_owner(golden, tmp_path, monkeypatch):
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    dst, sv_ids, counts = _case(golden, tmp_path)
    flt.write_handoff(dst, sv_ids, counts)
    assert flt.read_handoff(dst) is not None
    # same size, same time stamp, other content (what a coarse-mtime file system would let through): the digest catches it
    st = os.stat(dst)
    raw = bytearray(open(dst, "rb").read())
    i = raw.index(b"read")
    raw[i:i + 4] = b"READ"
    open(dst, "wb").write(raw)
    os.utime(dst, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert os.stat(dst).st_size == st.st_size and os.stat(dst).st_mtime_ns == st.st_mtime_ns
    assert flt.read_handoff(dst) is None
    # a table others could have written is not trusted
    flt.write_handoff(dst, sv_ids, counts)
    p = flt.handoff_path(dst)
    assert flt.read_handoff(dst) is not None
    os.chmod(p, 0o666)
    assert flt.read_handoff(dst) is None
    os.chmod(p, 0o600)
    os.chmod(os.path.dirname(p), 0o777)
    assert flt.read_handoff(dst) is None
    flt.write_handoff(dst, sv_ids, counts)                     # ... and nothing is written into such a directory either
    os.chmod(os.path.dirname(p), 0o700)


def test_error_order_of_a_file_that_is_not_utf8(golden):
    """svjg/filter.py: reference_error against golden/utf8order (the reference itself on GAFs that are not UTF-8 and hold a malformed
    line): it reads the file in text mode in blocks of 8192 bytes, so the UnicodeDecodeError of a block comes before the error of any
    malformed line that ends in or behind that block, and after the errors of earlier lines."""
    import base64
    cases = json.load(open(f"{golden}/utf8order/cases.json"))
    assert {c["error"] for c in cases.values()} == {"ValueError", "UnicodeDecodeError"}
    for name, c in cases.items():
        if c["bad_line_offset"] is None:
            continue
        data = np.frombuffer(base64.b64decode(c["gaf"]), dtype=np.uint8)
        e = ValueError("malformed GAF line")
        e.svjg_offset = c["bad_line_offset"]                    # (what libsvjg_hip reports through svjg_input_error)
        assert type(flt.reference_error(data, e)).__name__ == c["error"], name
    data = np.frombuffer(b"r\t1\t0\n" + b"x\t1\t2\n", dtype=np.uint8)   # pure ASCII: nothing changes
    e = ValueError("x")
    e.svjg_offset = 8
    assert flt.reference_error(data, e) is e


def test_gather_hits_fills_one_array():
    """filter.gather_hits: the records of several contexts land side by side in one array, in context order (stand-in contexts)."""
    import numpy as np
    from svjg import capi, filter as flt

    class Fake:
        def __init__(self, n, tag):
            self.n, self.tag = n, tag

        def stats(self):
            return {"n_hitrecs": self.n}

        def hits(self, out=None):
            if out is None:
                out = np.empty(self.n, dtype=capi.HITREC_DT)
            assert len(out) == self.n and out.flags["C_CONTIGUOUS"]
            out["line_start"] = np.arange(self.n) + 1000 * self.tag
            out["slot"] = self.tag
            return out
    ctxs = [Fake(5, 1), Fake(0, 2), Fake(7, 3)]
    r = flt.gather_hits(ctxs)
    assert len(r) == 12 and list(r["slot"]) == [1] * 5 + [3] * 7 and list(r["line_start"][:5]) == [1000 + i for i in range(5)] and r["line_start"][5] == 3000
    assert len(flt.gather_hits([Fake(4, 9)])) == 4

    class Broken(Fake):
        def hits(self, out=None):
            raise RuntimeError("copy failed")
    import pytest
    with pytest.raises(RuntimeError):
        flt.gather_hits([Fake(2, 1), Broken(3, 2)])
