#!/usr/bin/env python3
"""Generate tests/golden/fuzz/fuzz.json by RUNNING THE REFERENCE ITSELF on mutated GAF lines.

Runs only in the build container (needs /root/reference).  Every case is a small GAF fragment (a line of
tests/golden/testdir/test.gaf or of the quirk files after 1-3 random edits) evaluated by the reference's
filter-alignments.py against the reference's own test graph (tests/golden/testdir/test.gfa, test_svs_edges.json).
The fixture holds the fragment and what the reference did with it: the per-SV list lengths of its JSON, or the class
of the exception it died with.  Nothing of the reference's source is copied.

    python tests/golden/make_fuzz.py [n_cases] [seed]
"""
import base64
import json
import os
import random
import subprocess
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GRAPH = f"{HERE}/testdir"

NASTY = [b"\t", b" ", b"<", b">", b":", b"-", b".", b"0", b"9", b"x", b"\r", b"\x0b", b"_", b"+", b"@", b",", b"\n", b"\x1c"]
NUMS = [b"+5", b" 7", b"7 ", b"1_0", b"", b"-3", b"0", b"00", b"1e3", b"12345678901", b"9999999999", b"3.0", b"0x10", b"\xd9\xa3"]


TAGS = (b"\tcg:Z:10M2D", b"\tid:f:0.97", b"\tid:f:abc", b"cg:Z:", b"\tid:f:", b"\txx:Z:a>b<c")
# (tests/fuzz_big.py only: the committed fixture was drawn from TAGS)  identity tags of every shape, in front of and behind other tags, in the read name
MORE_TAGS = TAGS + (b"\tid:f:.5", b"\tid:f:7.", b"\tid:f:1e-3", b"\tid:f:0.5\tid:f:0.25", b"\tid:f:x\tid:f:1", b"\tid:f:1\tzd:Z:d:", b"\tid:f:0.123456789012345678901234567890123",
                    b"\tid:f:0..5", b"\tid:f:0.5 ", b"\tNM:i:3\tid:f:1.0\tdv:f:0.01", b"id:f:0.5", b"\tid:f", "\tid:f:\u0660".encode(), b"\tbd:i:3")


def mutate(line, rng, tags=TAGS):
    """1-3 edits of one GAF line (bytes, newline terminated)."""
    b = bytearray(line)
    for _ in range(rng.choice((1, 1, 1, 2, 2, 3))):
        cols = bytes(b).rstrip(b"\n").split(b"\t")
        op = rng.randrange(16)
        if op == 0 and len(b) > 1:                                  # replace a byte
            b[rng.randrange(len(b) - 1)] = rng.choice(NASTY)[0]
        elif op == 1 and len(b) > 1:                                # delete a byte
            del b[rng.randrange(len(b) - 1)]
        elif op == 2:                                               # insert a byte
            p = rng.randrange(len(b))
            b[p:p] = rng.choice(NASTY)
        elif op == 3 and len(cols) > 6:                             # a numeric column becomes something int() may or may not take
            c = rng.choice((1, 2, 3, 6, 7, 8, 9, 10, 11))
            if c < len(cols):
                cols[c] = rng.choice(NUMS)
                b = bytearray(b"\t".join(cols) + b"\n")
        elif op == 4 and len(cols) > 5:                             # revisit a node / repeat a step
            nodes = [x for x in cols[5].replace(b"<", b"\0<").replace(b">", b"\0>").split(b"\0") if x]
            if nodes:
                i = rng.randrange(len(nodes))
                nodes.insert(rng.randrange(len(nodes) + 1), nodes[i] if rng.random() < 0.5 else (b"<" if nodes[i][:1] == b">" else b">") + nodes[i][1:])
                cols[5] = b"".join(nodes)
                b = bytearray(b"\t".join(cols) + b"\n")
        elif op == 5 and len(cols) > 5:                             # flip an orientation
            ps = [i for i, ch in enumerate(cols[5]) if ch in b"<>"]
            if ps:
                p = rng.choice(ps)
                cols[5] = cols[5][:p] + (b"<" if cols[5][p:p + 1] == b">" else b">") + cols[5][p + 1:]
                b = bytearray(b"\t".join(cols) + b"\n")
        elif op == 6 and len(cols) > 2:                             # drop a column
            del cols[rng.randrange(len(cols))]
            b = bytearray(b"\t".join(cols) + b"\n")
        elif op == 7:                                               # tags the reference looks at
            b = bytearray(bytes(b).rstrip(b"\n") + rng.choice(tags) + b"\n")
        elif op == 8:                                               # trailing blanks / another terminator
            b = bytearray(bytes(b).rstrip(b"\n") + rng.choice((b" \n", b"\t\n", b"\r\n", b"\r", b"", b" \t \n", b"\n\n", b"\x0c\n")))
        elif op == 9 and len(cols) > 8:                             # alignment coordinates around the 100 bp rule
            c = rng.choice((6, 7, 8))
            try:
                cols[c] = str(max(0, int(cols[c]) + rng.choice((-150, -100, -99, -1, 1, 99, 100, 150, 5000)))).encode()
                b = bytearray(b"\t".join(cols) + b"\n")
            except ValueError:
                pass
        elif op == 10 and len(cols) > 5:                            # a node that is not in the graph / an odd spelling
            cols[5] = cols[5].replace(rng.choice((b"1:", b"2:", b"3:", b"-", b".")), rng.choice((b"1:0", b"9:", b"--", b"..", b":", b"")), 1)
            b = bytearray(b"\t".join(cols) + b"\n")
        elif op == 11:                                              # a read name with marks, quotes, non-ASCII
            cols[0] = rng.choice((b"a>b", b"<x", b'q"uo\\te', "réad".encode(), b"na\xffme", b"", b"cg:Z:"))
            b = bytearray(b"\t".join(cols) + b"\n")
        elif op == 12 and len(cols) > 5:                            # unoriented / single-node / empty paths
            cols[5] = rng.choice((cols[5][1:], cols[5].replace(b">", b",").replace(b"<", b","), b"", b">", cols[5].split(b">")[-1], b"*"))
            b = bytearray(b"\t".join(cols) + b"\n")
        elif op == 13 and len(cols) > 5:                            # a very long path (beyond the main kernel's node cap)
            cols[5] = cols[5] * rng.choice((2, 9, 40))
            b = bytearray(b"\t".join(cols) + b"\n")
        elif op == 14:                                              # two lines glued / split
            p = rng.randrange(len(b))
            b[p:p] = b"\n"
        else:                                                       # double tab
            p = bytes(b).find(b"\t", rng.randrange(len(b)))
            if p >= 0:
                b[p:p] = b"\t"
    return bytes(b)


def run_reference(fragment, tmp):
    gaf = os.path.join(tmp, "f.gaf")
    pre = os.path.join(tmp, "test")
    open(gaf, "wb").write(fragment)
    js = pre + "_informative_aln.json"
    if os.path.exists(js):
        os.remove(js)
    p = subprocess.run([sys.executable, f"{REF}/filter-alignments.py", "-a", gaf, "-g", pre + ".gfa", "-p", pre], capture_output=True, text=True)
    if p.returncode == 0:
        d = json.load(open(js))
        return {"rc": 0, "counts": {k: [len(v[0]), len(v[1])] for k, v in d.items()}}
    last = p.stderr.strip().splitlines()[-1] if p.stderr.strip() else ""
    return {"rc": p.returncode, "error": last.split(":")[0].split(".")[-1]}


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261003
    rng = random.Random(seed)
    base = open(f"{GRAPH}/test.gaf", "rb").read().splitlines(keepends=True)
    for f in sorted(os.listdir(f"{HERE}/quirks")):
        if f.endswith(".gaf") and not f.startswith("err_"):
            base += [l for l in open(f"{HERE}/quirks/{f}", "rb").read().splitlines(keepends=True) if l.count(b"\t") >= 11][:3]
    tmp = tempfile.mkdtemp()
    for ext in (".gfa", "_svs_edges.json"):
        os.symlink(f"{GRAPH}/test{ext}", os.path.join(tmp, "test" + ext))
    cases = []
    seen = set()
    while len(cases) < n_cases:
        frag = mutate(rng.choice(base), rng)
        if frag in seen:
            continue
        seen.add(frag)
        out = run_reference(frag, tmp)
        out["gaf"] = base64.b64encode(frag).decode()
        cases.append(out)
    os.makedirs(f"{HERE}/fuzz", exist_ok=True)
    json.dump({"seed": seed, "graph": "testdir/test.gfa + testdir/test_svs_edges.json", "cases": cases}, open(f"{HERE}/fuzz/fuzz.json", "w"), indent=0)
    n_ok = sum(c["rc"] == 0 for c in cases)
    errs = {}
    for c in cases:
        if c["rc"]:
            errs[c["error"]] = errs.get(c["error"], 0) + 1
    print(f"{len(cases)} cases: {n_ok} accepted ({sum(bool(c['counts']) for c in cases if c['rc'] == 0)} with hits), died: {errs}")


if __name__ == "__main__":
    main()
