"""The counts hand-off between the two drop-in scripts (svjg/filter.py: write_handoff / read_handoff): it must describe
exactly what a parse of the JSON gives, and must be ignored as soon as the JSON file is not the one it was written for."""
import json
import os
import shutil

import numpy as np

from svjg import capi, filter as flt


def test_handoff_matches_the_json(golden, tmp_path, monkeypatch):
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    tempfile.tempdir = None                                   # re-read TMPDIR
    try:
        src = f"{golden}/testdir/ref_informative_aln.json"
        dst = str(tmp_path / "p_informative_aln.json")
        shutil.copy(src, dst)
        ref = json.load(open(dst))
        sv_ids = sorted(ref) + ["chrZ:DEL-1-2"]               # one SV without informative alignments: not in the JSON
        counts = np.array([[len(ref[k][0]), len(ref[k][1])] for k in sorted(ref)] + [[0, 0]], dtype=np.uint32)
        perm = np.random.default_rng(1).permutation(len(sv_ids))    # slots are in graph order, not in key order
        flt.write_handoff(dst, [sv_ids[i] for i in perm], counts[perm])
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
    finally:
        tempfile.tempdir = None
