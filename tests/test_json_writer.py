"""The native _informative_aln.json writer (libsvjg_host.so, product code) against the reference's own JSON files.
No GPU needed: the hit records fed to it here come from the CPU oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle_c as OC
from oracle import oracle_py as O
from svjg import capi

QUIRKS = sorted(f[:-4] for f in os.listdir(os.path.join(os.path.dirname(__file__), "golden", "quirks"))
                if f.endswith(".gaf"))


def _recs_from_oracle(hits):
    recs = np.zeros(len(hits), dtype=capi.HITREC_DT)
    recs["line_start"] = hits["line_start"]
    recs["slot"] = hits["sv"]
    recs["n_ref"] = (hits["allele"] == 0).astype(np.uint16)
    recs["n_alt"] = (hits["allele"] == 1).astype(np.uint16)
    return recs


@pytest.mark.parametrize("name", QUIRKS)
def test_quirks(golden, name, tmp_path):
    q = f"{golden}/quirks"
    if json.load(open(f"{q}/manifest.json"))[name]["rc"]:
        pytest.skip("the reference dies on this input")
    orc = OC.COracle(O.load_edges(f"{q}/q_svs_edges.json"), O.load_alt_node_len(f"{q}/q.gfa"))
    raw = open(f"{q}/{name}.gaf", "rb").read()
    _, hits, _ = orc.filter(raw)
    out = str(tmp_path / "o.json")
    rng = np.random.default_rng(1)
    recs = _recs_from_oracle(hits)
    capi.write_informative_json(out, np.frombuffer(raw, dtype=np.uint8), recs[rng.permutation(len(recs))], orc.sv_ids, n_threads=3)
    assert open(out).read() == open(f"{q}/{name}.ref.json").read()


def test_testdir(golden, tmp_path):
    t = f"{golden}/testdir"
    orc = OC.COracle(O.load_edges(f"{t}/test_svs_edges.json"), O.load_alt_node_len(f"{t}/test.gfa"))
    raw = open(f"{t}/test.gaf", "rb").read()
    _, hits, _ = orc.filter(raw)
    out = str(tmp_path / "o.json")
    capi.write_informative_json(out, np.frombuffer(raw, dtype=np.uint8), _recs_from_oracle(hits), orc.sv_ids)
    assert open(out).read() == open(f"{t}/ref_informative_aln.json").read()


def test_synth_sha(golden, tmp_path):
    import synth
    g6 = json.load(open(f"{golden}/synth/g6.json"))["g6_mixed"]
    pre = str(tmp_path / "s")
    synth.generate(prefix=pre, **g6["args"])
    orc = OC.COracle(O.load_edges(pre + "_svs_edges.json"), O.load_alt_node_len(pre + ".gfa"))
    raw = np.fromfile(pre + ".gaf", dtype=np.uint8)
    _, hits, _ = orc.filter(raw)
    out = str(tmp_path / "o.json")
    capi.write_informative_json(out, raw, _recs_from_oracle(hits), orc.sv_ids, n_threads=8)
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == g6["sha256_json"]


def test_repeated_and_many_records(tmp_path):
    """a record that stands for several list elements (n_ref / n_alt > 1), lists of thousands of lines over several runs of
    keys, the last line without a terminator: the file is what json.dumps makes of the same dictionary"""
    lines = [f"read{i}\t{i}\tq\"\\\x01\u00e9\tcol\n".encode("utf-8") for i in range(3000)]
    lines[-1] = lines[-1][:-1]
    raw = b"".join(lines)
    starts = np.cumsum([0] + [len(x) for x in lines[:-1]])
    rng = np.random.default_rng(3)
    n_slots = 40
    recs = np.zeros(60000, dtype=capi.HITREC_DT)
    recs["line_start"] = starts[rng.integers(0, len(lines), len(recs))]
    recs["slot"] = rng.integers(0, n_slots - 3, len(recs))
    recs["n_ref"] = rng.integers(0, 3, len(recs))
    recs["n_alt"] = np.where(recs["n_ref"] == 0, rng.integers(1, 4, len(recs)), 0)
    ids = [f"chr{i % 7}:DEL-{i}-{i + 60}" for i in range(n_slots)]
    text = raw.decode("utf-8").splitlines(True)
    at = {int(s): t for s, t in zip(starts, text)}
    want = {}
    order = np.lexsort((np.arange(len(recs)), recs["line_start"]))          # file order inside a key
    for r in recs[order]:
        e = want.setdefault(ids[int(r["slot"])], [[], []])
        e[0] += [at[int(r["line_start"])]] * int(r["n_ref"]); e[1] += [at[int(r["line_start"])]] * int(r["n_alt"])
    out = str(tmp_path / "o.json")
    capi.write_informative_json(out, np.frombuffer(raw, dtype=np.uint8), recs, ids, n_threads=5)
    assert open(out).read() == json.dumps(want, sort_keys=True, indent=4)


def test_invalid_utf8(tmp_path):
    recs = np.zeros(1, dtype=capi.HITREC_DT)
    recs["n_ref"] = 1
    with pytest.raises(UnicodeDecodeError):
        capi.write_informative_json(str(tmp_path / "o.json"), np.frombuffer(b"r\xff\tx\n", dtype=np.uint8), recs, ["1:DEL-1-2"])


def test_count_reader_matches_json_load(golden, tmp_path):
    files = [f"{golden}/testdir/ref_informative_aln.json", f"{golden}/vcf/cases_informative_aln.json",
             f"{golden}/quirks/json_escapes.ref.json", f"{golden}/quirks/empty_file.ref.json", f"{golden}/quirks/repeat_flipped.ref.json"]
    odd = tmp_path / "odd.json"     # other valid spellings of the same shape: compact, extra elements, unicode key, nested values
    odd.write_text('{"a\\u00e9\\ud83d\\ude00":[["x",1,{"k":[1,2]}],[],"ignored"],\n "b" : [ [ ] , [null, true, 3.5e1] ] }')
    files.append(str(odd))
    for f in files:
        keys, cnt = capi.count_informative_json(f)
        d = json.load(open(f))
        assert keys == list(d)
        assert cnt.tolist() == [[len(d[k][0]), len(d[k][1])] for k in keys]


@pytest.mark.parametrize("text", ['{"a": [[1], [2]', '[1, 2]', '{"a": [[1]]}', '{"a": 5}', '{"a": [[1],[2]]} x', ''])
def test_count_reader_rejects(tmp_path, text):
    p = tmp_path / "bad.json"
    p.write_text(text)
    with pytest.raises(ValueError):
        capi.count_informative_json(str(p))


def _canonical(n_keys, rng, escapes=False):
    d = {}
    for k in range(n_keys):
        def text(i):
            s = f"read{k}_{i}\t20015\t12\t+\t>chr1:1000-2000<chr2:1-9\t300\ttp:A:P\n"
            return s + ('\n    "trap": [\\ "' if escapes and i % 7 == 3 else "")      # (escaped in the file: no raw newline inside a string)
        d[f"chr{k % 5}:DEL-{k * 1000}-{k * 1000 + 300}"] = [[text(i) for i in range(int(rng.integers(0, 40)))], [text(i) for i in range(int(rng.integers(0, 9)))]]
    return d


@pytest.mark.parametrize("threads", [2, 3, 8, 32])
def test_count_reader_in_parallel_equals_json_load(tmp_path, monkeypatch, threads):
    """r06: files of 64 MB and more are parsed by several threads, chained (svjg_json.cpp: count_chained); here small files are sent the
    same way.  json.dumps' own layout (the chain holds), and layouts whose guessed starts are wrong or absent (the chain fails, one thread
    parses): always what json.load says."""
    rng = np.random.default_rng(threads)
    monkeypatch.setenv("SVJG_JSON_PARALLEL_FROM", "0")
    monkeypatch.setenv("SVJG_JSON_THREADS", str(threads))
    d = _canonical(400, rng, escapes=True)
    layouts = {"indent4": json.dumps(d, sort_keys=True, indent=4), "indent2": json.dumps(d, indent=2), "compact": json.dumps(d, separators=(",", ":")),
               "indent4_crlf": json.dumps(d, indent=4).replace("\n", "\r\n"), "one_key": json.dumps({"k": [["a"] * 500, []]}, indent=4),
               "empty": "{}", "empty_lists": json.dumps({f"k{i}": [[], []] for i in range(300)}, indent=4)}
    # elements at the indent json.dumps gives keys: every guessed start is inside a pair
    layouts["elements_at_four"] = "{\n" + ",\n".join('"%s": [[\n    "x",\n    "y"\n],\n[\n    "z"\n]]' % k for k in list(d)[:200]) + "\n}"
    # extra elements behind the two lists, nested values as elements
    layouts["extras"] = json.dumps({k: [v[0], v[1], "more", {"x": [1, 2]}] for k, v in list(d.items())[:100]}, indent=4)
    for name, text in layouts.items():
        p = tmp_path / f"{name}.json"
        p.write_text(text)
        keys, cnt = capi.count_informative_json(str(p))
        want = json.loads(text)
        assert keys == list(want), name
        assert cnt.tolist() == [[len(want[k][0]), len(want[k][1])] for k in keys], name


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("text", ['{"a": [["x\ny"], []]}', '{"a": [["x\\qy"], []]}', '{"a": [["\\u12G4"], []]}', '{"a\tb": [[], []]}', '{"a": [["x\x01"], []]}',
                                  '{\n    "a": [[], []],\n    "b": [["x"], [] ],\n    "c": [[], []]\n    "d": [[], []]\n}',
                                  '{\n    "a": [[], []],\n    "b": [["x"], []]\n}\n    "c": [[], []]\n}'])
def test_count_reader_is_as_strict_as_json_load(tmp_path, monkeypatch, text, threads):
    """what json.load refuses (predict-genotype.py:67-68 dies with JSONDecodeError, a ValueError): raw control characters inside a string,
    escapes JSON does not have, a missing comma between pairs, text behind the object — by one thread and by the chained threads"""
    monkeypatch.setenv("SVJG_JSON_PARALLEL_FROM", "0")
    monkeypatch.setenv("SVJG_JSON_THREADS", str(threads))
    with pytest.raises(ValueError):
        json.loads(text)
    p = tmp_path / "bad.json"
    p.write_text(text)
    with pytest.raises(ValueError):
        capi.count_informative_json(str(p))


def test_count_reader_takes_and_refuses_the_values_json_load_does(tmp_path):
    """r06: list elements and extras that are neither strings nor containers are read with json.load's own grammar (true / false / null, NaN / Infinity /
    -Infinity, numbers); what it refuses is refused (the reference dies with JSONDecodeError there, predict-genotype.py:67-68)"""
    good = ['{"a": [[1, -0, 2.5e-3, 1E5, true, false, null], [NaN, Infinity, -Infinity]], "b": [[], [0.5]]}',
            '{"a": [[{"k": [1, {"z": null}]}, [[[]]]], [], 7, "x"]}', ' {\n"a"\t:\r[ [ ] , [ ] ] }\n ']
    bad = ['{"a": [[01], []]}', '{"a": [[1.], []]}', '{"a": [[.5], []]}', '{"a": [[+1], []]}', '{"a": [[1e], []]}', '{"a": [[tru], []]}', '{"a": [[True], []]}',
           '{"a": [[nan], []]}', '{"a": [[1 2], []]}', '{"a": [[abc], []]}', '{"a": [[1,], []]}', '{"a": [[,1], []]}', '{"a": [[--1], []]}', '{"a": [[0x10], []]}',
           '{"a": [[1e5x], []]}', '{"a": [[' + "[" * 2000 + "]" * 2000 + '], []]}', "\ufeff" + '{"a": [[], []]}', '{"a": [[], []],}', "{'a': [[], []]}"]
    for i, text in enumerate(good):
        d = json.loads(text)
        p = tmp_path / f"g{i}.json"
        p.write_text(text)
        keys, cnt = capi.count_informative_json(str(p))
        assert keys == list(d) and cnt.tolist() == [[len(d[k][0]), len(d[k][1])] for k in keys], text
    for i, text in enumerate(bad):
        with pytest.raises((ValueError, RecursionError)):
            json.loads(text)
        p = tmp_path / f"b{i}.json"
        p.write_text(text, encoding="utf-8")
        with pytest.raises(ValueError):
            capi.count_informative_json(str(p))


def test_count_reader_validates_utf8_inside_strings(tmp_path):
    """the reference reads the JSON in text mode: a byte sequence that is not UTF-8 — anywhere, also inside a string it would only count — is a
    UnicodeDecodeError (a ValueError) before json.load sees it; well-formed UTF-8 is text like any other"""
    ok = '{"kéy\U0001F600": [["æøå ü 漢字 \U0001F9EC", "plain"], ["xé"]], "b": [[], []]}'.encode("utf-8")
    p = tmp_path / "ok.json"
    p.write_bytes(ok)
    keys, cnt = capi.count_informative_json(str(p))
    d = json.loads(ok.decode("utf-8"))
    assert keys == list(d) and cnt.tolist() == [[len(d[k][0]), len(d[k][1])] for k in keys]
    for i, badseq in enumerate((b"\xff", b"\xc0\xaf", b"\xe0\x80\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xc3", b"\xe2\x82", b"\x80", b"\xf8\x88\x80\x80\x80")):
        for where in (b'{"a": [["x%sy"], []]}', b'{"a%s": [["x"], []]}', b'{"a": [["x"], []]}%s'):
            raw = where % badseq
            with pytest.raises(ValueError):
                json.loads(raw.decode("utf-8"))
            q = tmp_path / f"bad{i}.json"
            q.write_bytes(raw)
            with pytest.raises(ValueError):
                capi.count_informative_json(str(q))
