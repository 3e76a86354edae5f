#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  Nothing of the reference's source is
copied: the fixtures are inputs + the reference's outputs (and two of its own test *data* files,
test.vcf / expected_genotype.vcf).  The GPU box never runs this script.

    python tests/golden/make_golden.py            # regenerate the small groups below
    python tests/golden/make_golden.py full       # BASELINE configs at full size (c2: minutes, c3: half an hour and 27 GB, c4slice: minutes)
    python tests/golden/make_golden.py full:c4slice

Fixture groups:
  testdir/   G1  graph of the reference's test-dir (edges JSON + GFA with ref sequences elided)
             G2  count-matched GAF for that graph: reference filter+genotyper reproduce the 40
                 data rows of expected_genotype.vcf byte for byte
  quirks/    G3  hand-made graph + GAF lines exercising the quirks of SURVEY Appendix D,
                 with the reference's JSON (or the exception class it dies with)
  lik/       G4  known answers of the reference's likelihood()
  vcf/       G5  VCF-parsing cases through the reference's decision_vcf()
  synth/     G6  medium synthetic case from tools/svjg_synth (inputs regenerated from the seed):
                 sha256 of the reference JSON/VCF + the full count vector
  utf8order/ G8  GAFs that are not UTF-8 and hold a malformed line: which exception the reference dies with
  longpath/  G9  tests/longpath_fuzz.py: long walks (65..216 nodes) with one late event on graphs of >= 2 000 nodes
  longtail/  G11 lines longer than 8 KB with one event in the tail at boundary positions (tests/longpath_fuzz.py: make_tail_case)
  blanks/    G12 (r06) every blank-like byte around every decimal column, an id:f: value, the line's end (tests/alphabet_fuzz.py: blank_cases)
  fuzz7/     G13 (r06) 12 000 mutants over the full 7-bit alphabet (tests/alphabet_fuzz.py: mutate7), each through the reference
  hg002shape/ G14 (r06) BASELINE configs[4]'s shape: the graph built by the reference's construct-graph.py, 200 k lines and the whole 4.65 M-line
                 block through its filter and genotyper
  contigs/   G10 GRCh38 analysis-set contig names (HLA-DRB1*15:03:01:01, chrUn_JTFH01001998v1_decoy, chr6_GL000250v2_alt, chrEBV)
  realshape/ G7  lines shaped like real minigraph output (read names, cg:Z: / ds:Z: tags, paths of up to 300 nodes,
                 UCSC contig names) on a 600-SV graph, with the reference's JSON and VCF
             full  c2_full.json / c3_full.json / c4slice_full.json: sha256 of the reference's JSON and VCF for the
                 BASELINE configurations at full size, with its run time (make_full)
"""
import hashlib
import importlib.util
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile
import contextlib

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_filter = _load("ref_filter", f"{REF}/filter-alignments.py")
ref_geno = _load("ref_geno", f"{REF}/predict-genotype.py")


def run_ref_filter(gaf, gfa, prefix, extra=()):
    """-> (returncode, last stderr line)"""
    p = subprocess.run([sys.executable, f"{REF}/filter-alignments.py", "-a", gaf, "-g", gfa, "-p", prefix, *extra],
                       capture_output=True, text=True)
    err = p.stderr.strip().splitlines()[-1] if p.stderr.strip() else ""
    return p.returncode, err


def run_ref_genotype(js, vcf, out, ms=3, err=None):
    cmd = [sys.executable, f"{REF}/predict-genotype.py", "-d", js, "-v", vcf, "-o", out, "--minsupport", str(ms)]
    if err is not None:
        cmd += ["-e", str(err)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    return p.returncode, p.stdout


# ----------------------------------------------------------------------------------------------
# G1 + G2
# ----------------------------------------------------------------------------------------------

def gaf_line(name, path_nodes, orients, node_len, ts=0, te_back=0, extra="tp:A:P\tcm:i:50\ts1:i:500\ts2:i:0\tdv:f:0.0100"):
    tlen = sum(node_len[n] for n in path_nodes)
    te = tlen - te_back
    path = "".join(o + n for o, n in zip(orients, path_nodes))
    qlen = te - ts
    cols = [name, str(qlen), "0", str(qlen), "+", path, str(tlen), str(ts), str(te), str(qlen), str(qlen), "60"]
    return "\t".join(cols) + ("\t" + extra if extra else "") + "\n"


def make_testdir():
    out = f"{HERE}/testdir"
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp()
    subprocess.run([sys.executable, f"{REF}/construct-graph.py", "-v", f"{REF}/test-dir/test.vcf",
                    "-r", f"{REF}/test-dir/reference_genome.fasta", "-o", f"{tmp}/test.gfa"], check=True)
    shutil.copy(f"{tmp}/test_svs_edges.json", f"{out}/test_svs_edges.json")
    shutil.copy(f"{REF}/test-dir/test.vcf", f"{out}/test.vcf")
    shutil.copy(f"{REF}/test-dir/expected_genotype.vcf", f"{out}/expected_genotype.vcf")
    # GFA with reference-node sequences elided (the filter only measures alt-node sequences)
    node_len, order = {}, {}
    with open(f"{tmp}/test.gfa") as fi, open(f"{out}/test.gfa", "w") as fo:
        for ln in fi:
            if ln.startswith("S"):
                _, name, seq = ln.rstrip("\n").split("\t")
                node_len[name] = len(seq)
                if "." in name.split(":")[-1]:
                    fo.write(ln)
                else:
                    fo.write(f"S\t{name}\t*\n")
                    order.setdefault(name.split(":")[0], []).append(name)
            elif ln.startswith("P"):
                c = ln.rstrip("\n").split("\t")
                fo.write("\t".join([c[0], c[1], c[2], "*"]) + "\n")
            else:
                fo.write(ln)
    edges = json.load(open(f"{out}/test_svs_edges.json"))

    # target raw counts implied by expected_genotype.vcf
    target = {}
    ins_seen = {}
    for ln in open(f"{out}/expected_genotype.vcf"):
        if ln.startswith("#"):
            continue
        c = ln.rstrip("\n").split("\t")
        info = c[7]
        svt = info.split("SVTYPE=")[1].split(";")[0]
        ad = c[9].split(":")[2].split(",")
        a0, a1 = float(ad[0]), float(ad[1])
        if svt == "DEL":
            key = f"{c[0]}:DEL-{c[1]}-{info.split('END=')[1].split(';')[0]}"
            raw = (int(round(a0 * 2)), int(a1))
        elif svt == "INS":
            ins_seen[c[1]] = ins_seen.get(c[1], 0) + 1
            key = f"{c[0]}:INS-{c[1]}-{ins_seen[c[1]]}"
            raw = (int(a0), int(round(a1 * 2)))
        elif svt == "INV":
            key = f"{c[0]}:INV-{c[1]}-{info.split('END=')[1].split(';')[0]}"
            raw = (int(a0), int(a1))
        else:
            alt = c[4]
            br = "[" if "[" in alt else "]"
            p = [x for x in alt.split(br) if x]
            key = f"{c[0]}:BND-{c[1]}{br}{p[1]}{br}" if ":" in p[1] else f"{c[0]}:BND-{br}{p[0]}{br}{c[1]}"
            raw = (int(a0), int(a1))
        target[key] = raw

    # candidate path templates: each edge, extended along the reference path until both flanks >= 100 bp
    def ref_neighbour(node, direction):
        chrom = node.split(":")[0]
        if "." in node.split(":")[-1]:
            return None
        lst = order[chrom]
        i = lst.index(node) + direction
        return lst[i] if 0 <= i < len(lst) else None

    templates = []
    for key in edges:
        ln_, ls, rn, rs = key.split("@")
        nodes = [ln_, rn]
        ori = [">" if ls == "+" else "<", ">" if rs == "+" else "<"]
        for _ in range(6):  # extend left flank
            if sum(node_len[n] for n in nodes[:1]) >= 100 and len(nodes) >= 2:
                break
            nb = ref_neighbour(nodes[0], -1 if ori[0] == ">" else +1)
            if nb is None:
                break
            nodes.insert(0, nb)
            ori.insert(0, ori[0])
        for _ in range(6):  # extend right flank
            if node_len[nodes[-1]] >= 100:
                break
            nb = ref_neighbour(nodes[-1], +1 if ori[-1] == ">" else -1)
            if nb is None:
                break
            nodes.append(nb)
            ori.append(ori[-1])
        templates.append((nodes, ori))
        # reverse-strand twin of the same walk
        flip = {">": "<", "<": ">"}
        templates.append((nodes[::-1], [flip[o] for o in ori[::-1]]))

    # contribution vector of each template, measured with the reference filter itself
    tdir = tempfile.mkdtemp()
    with open(f"{tdir}/t.gaf", "w") as fh:
        for i, (nodes, ori) in enumerate(templates):
            fh.write(gaf_line(f"tmpl{i}", nodes, ori, node_len))
    shutil.copy(f"{out}/test_svs_edges.json", f"{tdir}/t_svs_edges.json")
    rc, err = run_ref_filter(f"{tdir}/t.gaf", f"{out}/test.gfa", f"{tdir}/t")
    assert rc == 0, err
    contrib = json.load(open(f"{tdir}/t_informative_aln.json"))
    keys = sorted(set(contrib) | set(target))
    rows = [(k, a) for k in keys for a in (0, 1)]
    A = np.zeros((len(rows), len(templates)))
    for ri, (k, a) in enumerate(rows):
        for txt in contrib.get(k, [[], []])[a]:
            A[ri, int(txt.split("\t")[0][4:])] += 1
    b = np.array([target.get(k, (None, None))[a] if k in target else -1 for k, a in rows], dtype=float)

    from scipy.optimize import milp, LinearConstraint, Bounds
    constrained = b >= 0          # orphan BND-mate keys are unconstrained
    res = milp(c=np.ones(len(templates)), integrality=np.ones(len(templates)),
               bounds=Bounds(0, np.inf),
               constraints=LinearConstraint(A[constrained], b[constrained], b[constrained]))
    assert res.success, res.message
    x = np.round(res.x).astype(int)

    rng = np.random.default_rng(20260515)
    lines = []
    rid = 0
    for ti, n in enumerate(x):
        nodes, ori = templates[ti]
        for _ in range(n):
            # jitter Ts / Te inside the slack so the 100-bp rule is exercised but still passes
            slack_l = node_len[nodes[0]] - 100 if len(nodes) == 2 else 0
            slack_r = node_len[nodes[-1]] - 100 if len(nodes) == 2 else 0
            ts = int(rng.integers(0, max(1, min(slack_l, 300) + 1)))
            teb = int(rng.integers(0, max(1, min(slack_r, 300) + 1)))
            lines.append(gaf_line(f"read{rid}", nodes, ori, node_len, ts=ts, te_back=teb))
            rid += 1
    # single-node alignments and one too-short overlap per template: must not change any count
    for ti in range(0, len(templates), 7):
        nodes, ori = templates[ti]
        lines.append(gaf_line(f"read{rid}", nodes[:1], ori[:1], node_len)); rid += 1
        if len(nodes) == 2 and node_len[nodes[0]] > 150:
            lines.append(gaf_line(f"read{rid}", nodes, ori, node_len, ts=node_len[nodes[0]] - 99)); rid += 1
    order_ix = rng.permutation(len(lines))
    with open(f"{out}/test.gaf", "w") as fh:
        for i in order_ix:
            fh.write(lines[i])

    work = tempfile.mkdtemp()
    for f in ("test.gaf", "test.gfa", "test_svs_edges.json"):
        shutil.copy(f"{out}/{f}", f"{work}/{f}")
    rc, err = run_ref_filter(f"{work}/test.gaf", f"{work}/test.gfa", f"{work}/test")
    assert rc == 0, err
    rc, so = run_ref_genotype(f"{work}/test_informative_aln.json", f"{out}/test.vcf", f"{work}/test_genotype.vcf")
    assert rc == 0
    got = [l for l in open(f"{work}/test_genotype.vcf") if not l.startswith("#")]
    exp = [l for l in open(f"{out}/expected_genotype.vcf") if not l.startswith("#")]
    assert got == exp, "count-matched GAF does not reproduce expected_genotype.vcf"
    shutil.copy(f"{work}/test_informative_aln.json", f"{out}/ref_informative_aln.json")
    shutil.copy(f"{work}/test_genotype.vcf", f"{out}/ref_genotype.vcf")
    with open(f"{out}/ref_stdout.txt", "w") as fh:
        fh.write(so)
    print(f"testdir: {len(lines)} alignments, {len(templates)} templates, 40/40 expected rows reproduced")


# ----------------------------------------------------------------------------------------------
# G3 quirks
# ----------------------------------------------------------------------------------------------

def make_quirks():
    out = f"{HERE}/quirks"
    os.makedirs(out, exist_ok=True)
    # Hand-made graph (not from construct-graph.py) so that hazard names can coexist.
    # chromosomes "1" and "11" (substring hazard Q6), alt nodes .1 / .10 at one position.
    nodes = {
        "1:1-1000": 1000, "1:1001-1500": 500, "1:1501-3000": 1500, "1:3001-3040": 40, "1:3041-5000": 1960,
        "11:1-1000": 1000, "11:1001-2000": 1000, "11:2001-4000": 2000,
        "1:1001.1": 250, "1:1001.10": 300, "1:3001.1": 120,
        "chrA:1-700": 700, "chrA:701-900": 200, "chrA:901-2000": 1100,
    }
    E = {}

    def add(l, ls, r, rs, sv, a):
        E.setdefault("@".join((l, ls, r, rs)), []).append([sv, a])

    # DEL 1000-1500 on chr 1
    add("1:1-1000", "+", "1:1001-1500", "+", "1:DEL-1000-1500", 0)
    add("1:1001-1500", "+", "1:1501-3000", "+", "1:DEL-1000-1500", 0)
    add("1:1-1000", "+", "1:1501-3000", "+", "1:DEL-1000-1500", 1)
    # two INS at the same position 1000 (ids .1 and .10 to build the '.1' in '.10' hazard) sharing the ref link
    add("1:1-1000", "+", "1:1001-1500", "+", "1:INS-1000-1", 0)
    add("1:1-1000", "+", "1:1001-1500", "+", "1:INS-1000-10", 0)
    add("1:1-1000", "+", "1:1001.1", "+", "1:INS-1000-1", 1)
    add("1:1001.1", "+", "1:1001-1500", "+", "1:INS-1000-1", 1)
    add("1:1-1000", "+", "1:1001.10", "+", "1:INS-1000-10", 1)
    add("1:1001.10", "+", "1:1001-1500", "+", "1:INS-1000-10", 1)
    # short DEL 3000-3040 (40-bp inner node: flank sums needed)
    add("1:1501-3000", "+", "1:3001-3040", "+", "1:DEL-3000-3040", 0)
    add("1:3001-3040", "+", "1:3041-5000", "+", "1:DEL-3000-3040", 0)
    add("1:1501-3000", "+", "1:3041-5000", "+", "1:DEL-3000-3040", 1)
    # INS at 3000 (alt node 1:3001.1)
    add("1:1501-3000", "+", "1:3001-3040", "+", "1:INS-3000-1", 0)
    add("1:1501-3000", "+", "1:3001.1", "+", "1:INS-3000-1", 1)
    add("1:3001.1", "+", "1:3001-3040", "+", "1:INS-3000-1", 1)
    # chr 11: INV 1000-2000
    add("11:1-1000", "+", "11:1001-2000", "+", "11:INV-1000-2000", 0)
    add("11:1001-2000", "+", "11:2001-4000", "+", "11:INV-1000-2000", 0)
    add("11:1-1000", "+", "11:1001-2000", "-", "11:INV-1000-2000", 1)
    add("11:1001-2000", "-", "11:2001-4000", "+", "11:INV-1000-2000", 1)
    # BND 1 -> 11 with orphan mate key and a key stored in "reverse" form only
    add("1:3041-5000", "+", "11:2001-4000", "-", "1:BND-5000]11:4000]", 1)
    add("11:2001-4000", "-", "11:1001-2000", "-", "11:BND-5000]11:4000]", 0)
    # palindromic link: key equals its own reverse -> processed twice
    add("chrA:701-900", "+", "chrA:701-900", "-", "chrA:INV-700-900", 1)
    add("chrA:1-700", "+", "chrA:701-900", "+", "chrA:INV-700-900", 0)
    add("chrA:701-900", "+", "chrA:901-2000", "+", "chrA:INV-700-900", 0)
    # fwd and rev keys both present with different payloads
    add("chrA:901-2000", "-", "chrA:701-900", "-", "chrA:DEL-700-900", 0)
    # same (sv, allele) listed twice under one key (multiplicity)
    add("chrA:1-700", "+", "chrA:901-2000", "+", "chrA:DEL-700-900", 1)
    add("chrA:1-700", "+", "chrA:901-2000", "+", "chrA:DEL-700-900", 1)

    with open(f"{out}/q_svs_edges.json", "w") as fh:
        fh.write(json.dumps(E, sort_keys=True, indent=4))
    with open(f"{out}/q.gfa", "w") as fh:
        for n, l in nodes.items():
            if "." in n.split(":")[-1]:
                fh.write(f"S\t{n}\t{'A' * l}\n")
            else:
                fh.write(f"S\t{n}\t*\n")

    L = nodes
    tags = "tp:A:P\tcm:i:9\ts1:i:90\ts2:i:0\tdv:f:0.0200"

    def g(name, nodes_, ori, **kw):
        return gaf_line(name, nodes_, list(ori), L, extra=kw.pop("extra", tags), **kw)

    cases = {}
    cases["plain_ref_del"] = [g("r0", ["1:1-1000", "1:1001-1500", "1:1501-3000"], ">>>")]
    cases["alt_del"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>")]
    cases["reverse_strand"] = [g("r0", ["1:1501-3000", "1:1001-1500", "1:1-1000"], "<<<"),
                               g("r1", ["1:1501-3000", "1:1-1000"], "<<")]
    cases["shared_ref_link_ins_pair"] = [g("r0", ["1:1-1000", "1:1001-1500"], ">>")]
    cases["ins_alt_1"] = [g("r0", ["1:1-1000", "1:1001.1", "1:1001-1500"], ">>>")]
    cases["ins_alt_10_then_1_hazard"] = [
        # '1:1001.1' is a substring of '1:1001.10' placed earlier in the path (Q6)
        g("r0", ["1:1001.10", "1:1001-1500", "1:1-1000", "1:1001.1", "1:1001-1500"], "<><>>"),
        g("r1", ["1:1-1000", "1:1001.10", "1:1001-1500"], ">>>"),
    ]
    cases["chrom_1_in_11_hazard"] = [
        # '1:1-1000' is a substring of '11:1-1000' placed earlier (Q6): strand read from the wrong char
        g("r0", ["11:1-1000", "1:1-1000", "1:1001-1500"], "<>>"),
        g("r1", ["11:1001-2000", "11:1-1000", "1:1-1000", "1:1501-3000"], "<<>>"),
    ]
    cases["repeat_flipped"] = [
        # a node visited twice with opposite orientations: first-occurrence strand/index (Q4/Q5)
        g("r0", ["1:1-1000", "1:1001-1500", "1:1-1000", "1:1501-3000"], "><<>"),
        g("r1", ["1:1001-1500", "1:1-1000", "1:1001-1500", "1:1501-3000"], "<<>>"),
        g("r2", ["1:1-1000", "1:1001-1500", "1:1501-3000", "1:1-1000", "1:1001-1500"], ">>>>>"),
    ]
    cases["short_inner_node"] = [
        g("r0", ["1:1501-3000", "1:3001-3040"], ">>"),                     # right flank 40 (+1) < 100
        g("r1", ["1:1501-3000", "1:3001-3040", "1:3041-5000"], ">>>"),
        g("r2", ["1:1501-3000", "1:3001.1", "1:3001-3040", "1:3041-5000"], ">>>>"),
    ]
    b = []
    for d in (98, 99, 100, 101):                                            # left flank boundary
        b.append(g(f"l{d}", ["1:1-1000", "1:1001-1500"], ">>", ts=1000 - d))
    for d in (98, 99, 100, 101):                                            # right flank: len - (Tlen-Te-1)
        b.append(g(f"r{d}", ["1:1-1000", "1:1001-1500"], ">>", te_back=500 + 1 - d))
    cases["dover_boundaries"] = b
    cases["single_node_and_unoriented"] = [
        g("r0", ["1:1-1000"], ">"),
        "r1\t500\t0\t500\t+\t1:1-1000\t1000\t0\t500\t500\t500\t60\t" + tags + "\n",
        g("r2", ["1:1-1000", "1:1001-1500"], ">>"),
    ]
    cases["inv_paths"] = [
        g("r0", ["11:1-1000", "11:1001-2000", "11:2001-4000"], "><>"),
        g("r1", ["11:2001-4000", "11:1001-2000", "11:1-1000"], "<><"),
        g("r2", ["11:1-1000", "11:1001-2000", "11:2001-4000"], ">>>"),
    ]
    cases["bnd_and_orphan"] = [
        g("r0", ["1:3041-5000", "11:2001-4000", "11:1001-2000"], "><<"),
        g("r1", ["11:1001-2000", "11:2001-4000", "1:3041-5000"], ">><"),
    ]
    cases["palindromic_link"] = [
        g("r0", ["chrA:1-700", "chrA:701-900", "chrA:701-900", "chrA:901-2000"], ">><>"),
        g("r1", ["chrA:701-900", "chrA:701-900"], "><"),
    ]
    cases["fwd_and_rev_keys"] = [
        g("r0", ["chrA:701-900", "chrA:901-2000"], ">>"),
        g("r1", ["chrA:901-2000", "chrA:701-900"], "<<"),
        g("r2", ["chrA:1-700", "chrA:901-2000"], ">>"),
    ]
    cases["cg_tag"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>", extra=tags + "\tcg:Z:100M5D200M"),
                       g("r1", ["1:1-1000", "1:1501-3000"], ">>", extra="cg:Z:300M\t" + tags)]
    cases["id_tag"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>", extra="NM:i:3\tid:f:0.987\t" + tags)]
    # id:f: tags of every shape next to one another (the value is float()ed, Alen is not divided by; the LAST tag of a line counts)
    z = "r{}\t500\t0\t500\t+\t>1:1-1000>1:1001-1500\t1500\t0\t1500\t0\t0\t60\t"          # Alen == 0 lines: fine only with a tag
    cases["id_tag_forms"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>", extra=tags + "\tid:f:1"),
                             g("r1", ["1:1-1000", "1:1501-3000"], ">>", extra="id:f:.5\t" + tags),
                             g("r2", ["1:1-1000", "1:1001-1500"], ">>", extra=tags + "\tid:f:7."),
                             z.format(3) + "id:f:0.000001\t" + tags + "\n",
                             z.format(4) + tags + "\tid:f:12345678.25\n",
                             g("r5", ["1:1-1000", "1:1501-3000"], ">>", extra="id:f:junk\t" + tags + "\tid:f:0.75"),        # two tags: the last one counts
                             z.format(6) + "id:f:1e-3\n", z.format(7) + "id:f:+0.5\n", z.format(8) + "id:f:nan\n", z.format(9) + "id:f: 0.25\n",
                             g("sd:3_r10", ["1:1-1000", "1:1501-3000"], ">>", extra=tags),                                   # "d:" in the read name only: no tag
                             g("xid:f:0.5", ["1:1-1000", "1:1501-3000"], ">>", extra=tags).replace("xid:f:0.5\t", "xid:f:0.5\t", 1)]   # the tag IS the read name's tail
    cases["err_id_tag_in_read_name"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>"), g("id:f:0.5/ccs", ["1:1-1000", "1:1501-3000"], ">>", extra=tags)]
    cases["err_id_tag_empty"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>", extra=tags + "\tid:f:")]
    cases["err_id_tag_two_dots"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>", extra="id:f:1.2.3\t" + tags)]
    cases["err_id_tag_last_is_junk"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>", extra="id:f:0.5\t" + tags + "\tid:f:x")]
    cases["err_zero_alen_behind_tagged_lines"] = [z.format(0) + "id:f:0.5\n", z.format(1) + tags + "\n"]
    cases["json_escapes"] = [g('r"q\\x', ["1:1-1000", "1:1501-3000"], ">>"),
                             g("ré中", ["1:1-1000", "1:1501-3000"], ">>"),
                             g("r\x01\x7f", ["1:1-1000", "1:1501-3000"], ">>")]
    cases["no_trailing_newline"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>"),
                                    g("r1", ["1:1-1000", "1:1501-3000"], ">>").rstrip("\n")]
    cases["crlf_and_trailing_ws"] = [g("r0", ["1:1-1000", "1:1501-3000"], ">>").rstrip("\n") + "\r\n",
                                     g("r1", ["1:1-1000", "1:1501-3000"], ">>").rstrip("\n") + " \t\n"]
    cases["unknown_nodes"] = [
        g("r0", ["1:1-1000", "1:1001-1500"], ">>").replace("1:1001-1500", "1:1001-1400"),   # name not in graph
        g("r1", ["1:1-1000", "1:1001-1500"], ">>").replace(">1:1-1000", ">9:1-1000"),
        g("r2", ["1:1-1000", "1:1001-1500"], ">>").replace("1:1001-1500", "1:01001-1500"),  # non-canonical digits
    ]
    cases["tlen_mismatch"] = [
        # Tlen column disagrees with the node-name sum: the reference trusts the names for the sums
        g("r0", ["1:1-1000", "1:1001-1500"], ">>").replace("\t1500\t0\t1500\t", "\t1400\t0\t1400\t"),
        g("r1", ["1:1-1000", "1:1001-1500"], ">>").replace("\t1500\t0\t1500\t", "\t1700\t0\t1500\t"),
    ]
    cases["empty_file"] = []
    cases["many_nodes"] = [g("r0", ["1:1-1000", "1:1001-1500", "1:1501-3000", "1:3001-3040", "1:3041-5000",
                                    "11:2001-4000", "11:1001-2000", "11:1-1000"], ">>>>><<<")]
    # inputs the reference dies on (exit code 1): recorded with the exception class
    cases["err_few_columns"] = ["r0\t10\t0\t10\t+\t>1:1-1000>1:1001-1500\t1500\t0\n"]
    cases["err_nonint"] = [g("r0", ["1:1-1000", "1:1001-1500"], ">>").replace("\t60\t", "\tx\t")]
    cases["err_empty_line"] = [g("r0", ["1:1-1000", "1:1001-1500"], ">>"), "\n"]
    cases["err_zero_alen"] = ["r0\t500\t0\t500\t+\t>1:1-1000>1:1001-1500\t1500\t0\t1500\t0\t0\t60\t" + tags + "\n"]
    cases["ok_zero_alen_with_id"] = ["r0\t500\t0\t500\t+\t>1:1-1000>1:1001-1500\t1500\t0\t1500\t0\t0\t60\tid:f:0.5\n"]
    cases["err_unknown_alt_node"] = [g("r0", ["1:1-1000", "1:1001-1500"], ">>").replace(">1:1-1000", ">1:7.1>1:1-1000")
                                     .replace("\t1500\t0\t1500\t", "\t1600\t0\t1600\t")]
    cases["ok_unknown_alt_node_no_hit"] = ["r0\t500\t0\t500\t+\t>1:7.1>1:1-1000\t1100\t0\t1100\t500\t500\t60\t" + tags + "\n"]
    cases["err_unoriented_two"] = ["r0\t500\t0\t500\t+\t1:1-1000+,1:1001-1500+\t1500\t0\t1500\t500\t500\t60\t" + tags + "\n"]
    cases["err_bad_ref_coords"] = ["r0\t500\t0\t500\t+\t>1:1-1000>1:1001-1500>1:abc\t1500\t0\t1500\t500\t500\t60\t" + tags + "\n"]

    manifest = {}
    tdir = tempfile.mkdtemp()
    shutil.copy(f"{out}/q_svs_edges.json", f"{tdir}/q_svs_edges.json")
    for name, lines in cases.items():
        gaf = f"{out}/{name}.gaf"
        with open(gaf, "w", encoding="utf-8") as fh:
            fh.write("".join(lines))
        js = f"{tdir}/q_informative_aln.json"
        if os.path.exists(js):
            os.remove(js)
        rc, err = run_ref_filter(gaf, f"{out}/q.gfa", f"{tdir}/q")
        if rc == 0:
            shutil.copy(js, f"{out}/{name}.ref.json")
            manifest[name] = {"rc": 0, "n_lines": len(lines)}
        else:
            assert rc == 1
            manifest[name] = {"rc": 1, "error": err.split(":")[0], "n_lines": len(lines)}
    with open(f"{out}/manifest.json", "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("quirks:", {k: (v["rc"], v.get("error", "")) for k, v in manifest.items() if v["rc"]},
          f"{len(manifest)} cases")


# ----------------------------------------------------------------------------------------------
# G3b lines whose decimal columns are written with non-ASCII digits / blanks (int(), float() and str.rstrip() take them)
# ----------------------------------------------------------------------------------------------

def make_unicode():
    """golden/unicode: the quirks graph; GAF lines with Arabic-Indic / fullwidth digits and Unicode blanks in the integer columns,
    at the line's end and in an id:f: value, alone and between ordinary lines; what the reference wrote or died with."""
    q = f"{HERE}/quirks"
    out = f"{HERE}/unicode"
    os.makedirs(out, exist_ok=True)
    L = {}
    for line in open(f"{q}/q.gfa"):
        c = line.rstrip("\n").split("\t")
        if c[0] == "S":
            L[c[1]] = len(c[2]) if "." in c[1].split(":")[-1] else int(c[1].split(":")[-1].split("-")[1]) - int(c[1].split(":")[-1].split("-")[0]) + 1
    tags = "tp:A:P\tcm:i:9\ts1:i:90\ts2:i:0\tdv:f:0.0200"

    def g(name, nodes_, ori, **kw):
        return gaf_line(name, nodes_, list(ori), L, extra=kw.pop("extra", tags), **kw)

    def arabic(s):
        return "".join(chr(0x660 + int(ch)) if ch.isdigit() else ch for ch in s)

    def wide(s):
        return "".join(chr(0xFF10 + int(ch)) if ch.isdigit() else ch for ch in s)

    def cols(line, f, which):
        c = line.rstrip("\n").split("\t")
        for i in which:
            c[i] = f(c[i])
        return "\t".join(c) + "\n"
    plain = g("r0", ["1:1-1000", "1:1001-1500", "1:1501-3000"], ">>>")
    alt = g("r1", ["1:1-1000", "1:1501-3000"], ">>")
    ins = g("r2", ["1:1-1000", "1:1001.1", "1:1001-1500"], ">>>")
    cases = {}
    cases["digits_in_every_column"] = [plain, cols(alt, arabic, (1, 2, 3, 6, 7, 8, 9, 10, 11)), cols(ins, wide, (6, 7, 8)), alt]
    cases["blanks"] = [cols(plain, lambda x: "\u2003" + x + "\u00a0", (7, 8)), alt.rstrip("\n") + "\u2003\u3000\n", cols(ins, lambda x: " " + arabic(x), (1, 6))]
    cases["mixed_and_signs"] = [cols(alt, lambda x: "+" + arabic(x)[:1] + x[1:], (6, 8)), cols(plain, lambda x: arabic(x[:1]) + "_" + x[1:] if len(x) > 1 else x, (6, 8))]
    cases["id_tag_value"] = ["r3\t500\t0\t500\t+\t>1:1-1000>1:1001-1500\t1500\t0\t1500\t0\t0\t60\tid:f:" + arabic("0.5") + "\n",
                             alt.rstrip("\n") + "\tid:f:" + wide("1") + "e" + arabic("2") + "\n"]
    cases["overlap_boundary"] = [cols(g(f"l{d}", ["1:1-1000", "1:1001-1500"], ">>", ts=1000 - d), arabic, (7,)) for d in (99, 100)]
    cases["err_not_a_digit"] = [plain, cols(alt, lambda x: x + "\u00e9", (7,)), alt]
    cases["err_behind_unicode_lines"] = [cols(alt, arabic, (6, 7, 8)), plain.replace("\t60\t", "\tx\t"), cols(ins, wide, (6,))]
    cases["err_unicode_zero_alen"] = ["r3\t500\t0\t500\t+\t>1:1-1000>1:1001-1500\t1500\t0\t1500\t0\t" + arabic("0") + "\t60\t" + tags + "\n"]
    cases["err_unicode_before_ascii_error"] = [cols(alt, lambda x: x + "\u0663x", (7,)), plain.replace("\t60\t", "\t\t")]
    cases["err_id_tag_value"] = [alt.rstrip("\n") + "\tid:f:" + arabic("0.5") + "\u00e9\n"]
    # (r04) an id:f: "tag" INSIDE the path column whose value is written with non-ASCII digits: float() takes it, Alen == 0 raises nothing,
    # the piece of path is a node name the graph does not have; and the same node inside an overlap sum (get_node_len dies on it)
    cases["id_tag_in_path_column"] = [plain, "r3\t500\t0\t500\t+\t>1:1-1000>1:1001-1500id:f:" + arabic("5") + "\t1500\t0\t1500\t0\t0\t60\t" + tags + "\n", alt]
    cases["err_id_tag_in_path_column_node_in_sum"] = [plain, "r3\t500\t0\t500\t+\t>1:1-1000>1:1001-1500>1:1501-3000id:f:" + wide("7.5") + "\t3000\t0\t3000\t500\t500\t60\t" + tags + "\n"]
    manifest = {}
    tdir = tempfile.mkdtemp()
    shutil.copy(f"{q}/q_svs_edges.json", f"{tdir}/q_svs_edges.json")
    for name, lines in cases.items():
        gaf = f"{out}/{name}.gaf"
        with open(gaf, "w", encoding="utf-8") as fh:
            fh.write("".join(lines))
        js = f"{tdir}/q_informative_aln.json"
        if os.path.exists(js):
            os.remove(js)
        rc, err = run_ref_filter(gaf, f"{q}/q.gfa", f"{tdir}/q")
        if rc == 0:
            shutil.copy(js, f"{out}/{name}.ref.json")
            manifest[name] = {"rc": 0, "n_lines": len(lines)}
        else:
            assert rc == 1
            manifest[name] = {"rc": 1, "error": err.split(":")[0], "n_lines": len(lines)}
    with open(f"{out}/manifest.json", "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("unicode:", {k: (v["rc"], v.get("error", "")) for k, v in manifest.items()})


def make_dover():
    """golden/dover (r04): the quirks graph through the reference WITH -O 50.  argparse leaves a list in d_over, so the reference
    dies with TypeError where it first compares an overlap with it — the first link that has a candidate SV, once the node lengths
    of its left sum are there — and not before: lines in front of it are classified (and may die) as usual, a GAF without such a link
    is written as `{}`."""
    q = f"{HERE}/quirks"
    out = f"{HERE}/dover"
    os.makedirs(out, exist_ok=True)
    L = {}
    for line in open(f"{q}/q.gfa"):
        c = line.rstrip("\n").split("\t")
        if c[0] == "S":
            L[c[1]] = len(c[2]) if "." in c[1].split(":")[-1] else int(c[1].split(":")[-1].split("-")[1]) - int(c[1].split(":")[-1].split("-")[0]) + 1
    tags = "tp:A:P\tcm:i:9\ts1:i:90\ts2:i:0\tdv:f:0.0200"

    def g(name, nodes_, ori, **kw):
        return gaf_line(name, nodes_, list(ori), L, extra=kw.pop("extra", tags), **kw)
    plain = g("r0", ["1:1-1000", "1:1001-1500", "1:1501-3000"], ">>>")
    alt = g("r1", ["1:1-1000", "1:1501-3000"], ">>")
    single = g("s0", ["1:1-1000"], ">")
    nolink = g("n0", ["chrA:901-2000", "1:1-1000"], ">>")                       # two nodes, no such link in the table
    unknown = g("u0", ["1:1-1000", "1:1001-1500"], ">>").replace("1:1001-1500", "1:1001-1400")   # a name the graph does not have
    nonint = plain.replace("\t60\t", "\tx\t")
    cases = {}
    cases["dover_flag_hit"] = [plain, alt]
    cases["dover_flag_nohit"] = [single, nolink, unknown, single]
    cases["dover_flag_empty"] = []
    cases["dover_flag_error_in_front_of_the_hit"] = [nolink, nonint, plain]
    cases["dover_flag_hit_in_front_of_the_error"] = [nolink, alt, nonint]
    cases["dover_flag_keyerror_in_the_left_sum"] = [g("r0", ["1:1-1000", "1:1001-1500"], ">>").replace(">1:1-1000", ">1:7.1>1:1-1000").replace("\t1500\t0\t1500\t", "\t1600\t0\t1600\t")]
    cases["dover_flag_right_sum_not_reached"] = ["r0\t500\t0\t500\t+\t>1:1-1000>1:1001-1500>1:abc\t1500\t0\t1500\t500\t500\t60\t" + tags + "\n"]
    cases["dover_flag_hit_behind_many_lines"] = [single, nolink, unknown] * 40 + [g("r9", ["1:1501-3000", "1:1-1000"], "<<")] + [plain] * 3
    manifest = {}
    tdir = tempfile.mkdtemp()
    shutil.copy(f"{q}/q_svs_edges.json", f"{tdir}/q_svs_edges.json")
    for name, lines in cases.items():
        gaf = f"{out}/{name}.gaf"
        with open(gaf, "w", encoding="utf-8") as fh:
            fh.write("".join(lines))
        js = f"{tdir}/q_informative_aln.json"
        if os.path.exists(js):
            os.remove(js)
        rc, err = run_ref_filter(gaf, f"{q}/q.gfa", f"{tdir}/q", extra=("-O", "50"))
        if rc == 0:
            shutil.copy(js, f"{out}/{name}.ref.json")
            manifest[name] = {"rc": 0, "n_lines": len(lines)}
        else:
            assert rc == 1
            manifest[name] = {"rc": 1, "error": err.split(":")[0], "message": err, "n_lines": len(lines)}
    with open(f"{out}/manifest.json", "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("dover:", {k: (v["rc"], v.get("error", "")) for k, v in manifest.items()})


def make_nosv():
    """golden/nosv: a VCF none of whose records makes the constructor add an SV (a DUP, a symbolic INS without a sequence, an INS
    with a long REF): the graph the reference builds from it has no link with an SV (`{}`), the filter writes `{}` whatever the
    alignments say, and predict-genotype.py answers ./. for every row."""
    out = f"{HERE}/nosv"
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp()
    fa = f"{REF}/test-dir/reference_genome.fasta"
    chrom = open(fa).readline()[1:].split()[0]
    hdr = [ln for ln in open(f"{REF}/test-dir/test.vcf") if ln.startswith("#")]
    rows = [f"{chrom}\t1000\tdup1\tN\t<DUP>\t.\tPASS\tSVTYPE=DUP;END=1400;SVLEN=400\n",
            f"{chrom}\t3000\tins_noseq\tN\t<INS>\t.\tPASS\tSVTYPE=INS;END=3000;SVLEN=120\n",
            f"{chrom}\t5000\tins_longref\tACGT\t{'ACGT' * 30}\t.\tPASS\tSVTYPE=INS;END=5000;SVLEN=116\n"]
    with open(f"{out}/nosv.vcf", "w") as fh:
        fh.write("".join(hdr) + "".join(rows))
    subprocess.run([sys.executable, f"{REF}/construct-graph.py", "-v", f"{out}/nosv.vcf", "-r", fa, "-o", f"{tmp}/nosv.gfa"], check=True)
    shutil.copy(f"{tmp}/nosv_svs_edges.json", f"{out}/nosv_svs_edges.json")
    names = []
    with open(f"{tmp}/nosv.gfa") as fi, open(f"{out}/nosv.gfa", "w") as fo:      # reference-node sequences elided, as in testdir
        for ln in fi:
            c = ln.rstrip("\n").split("\t")
            if c[0] == "S":
                names.append((c[1], len(c[2])))
                fo.write(f"S\t{c[1]}\t*\n")
            elif c[0] == "P":
                fo.write("\t".join([c[0], c[1], c[2], "*"]) + "\n")
            else:
                fo.write(ln)
    L = dict(names)
    first = names[0][0]
    lines = [gaf_line("r0", [first], [">"], L), gaf_line("r1", [first], ["<"], L, ts=50, te_back=20)]
    if len(names) > 1:
        lines.append(gaf_line("r2", [names[0][0], names[1][0]], [">", ">"], L))
    with open(f"{out}/nosv.gaf", "w") as fh:
        fh.write("".join(lines))
    rc, err = run_ref_filter(f"{out}/nosv.gaf", f"{out}/nosv.gfa", f"{tmp}/nosv")
    assert rc == 0, err
    shutil.copy(f"{tmp}/nosv_informative_aln.json", f"{out}/nosv.ref.json")
    rc, so = run_ref_genotype(f"{tmp}/nosv_informative_aln.json", f"{out}/nosv.vcf", f"{out}/nosv.ref_genotype.vcf")
    with open(f"{out}/manifest.json", "w") as fh:
        json.dump({"edges": json.load(open(f"{out}/nosv_svs_edges.json")), "json": open(f"{out}/nosv.ref.json").read(), "genotype_rc": rc,
                   "genotype_stdout": so, "nodes": len(names)}, fh, indent=1, sort_keys=True)
    print("nosv:", open(f"{out}/manifest.json").read()[:400])


# ----------------------------------------------------------------------------------------------
# G4 likelihood known answers
# ----------------------------------------------------------------------------------------------

def make_lik():
    out = f"{HERE}/lik"
    os.makedirs(out, exist_ok=True)
    types = ["DEL", "INS", "INV", "BND"]
    rows = []
    rng = np.random.default_rng(4)
    cases = [(t, a, b, ms, 5e-5) for t in range(4) for a in range(61) for b in range(61) for ms in (1, 3)]
    for _ in range(3000):
        cases.append((int(rng.integers(4)), int(rng.integers(0, 5001)), int(rng.integers(0, 5001)), 3, 5e-5))
    for _ in range(1500):
        cases.append((int(rng.integers(4)), int(rng.integers(0, 300)), int(rng.integers(0, 300)),
                      int(rng.integers(0, 8)), [1e-3, 1e-2, 5e-5][int(rng.integers(3))]))
    for _ in range(300):
        cases.append((int(rng.integers(4)), int(rng.integers(0, 60001)), int(rng.integers(0, 60001)), 3, 5e-5))
    gtc = {"0/0": 0, "0/1": 1, "1/1": 2, "./.": 3}
    for t, a, b, ms, e in cases:
        cnt = [a, b]
        gt, pl = ref_geno.likelihood(cnt, types[t], ms, e)
        dp = str(round(sum(cnt), 3))
        rows.append((t, a, b, ms, e, gtc[gt], int(pl[0]), int(pl[1]), int(pl[2]), f"{dp}:{cnt[0]},{cnt[1]}"))
    arr = np.array([r[:4] + r[5:9] for r in rows], dtype=np.int64)
    errs = np.array([r[4] for r in rows], dtype=np.float64)
    txt = np.array([r[9] for r in rows])
    np.savez_compressed(f"{out}/lik_kat.npz", cases=arr, err=errs, dp_ad=txt)
    print(f"lik: {len(rows)} known answers")


def make_lik_boundary():
    """lik/lik_boundary.npz: known answers of likelihood() where a PL lies next to an integer boundary (found by a search in
    double precision over counts up to 3 000, then run through the reference), and for deep samples (counts up to 10^6)."""
    import math
    out = f"{HERE}/lik"
    types = ["DEL", "INS", "INV", "BND"]
    e = 5e-5
    l_ok, l_err, l_half = math.log10(1 - e), math.log10(e), math.log10(0.5)
    lg = np.array([math.lgamma(i + 1) for i in range(6002)]) / math.log(10)
    a = np.arange(0, 3001, dtype=np.float64)
    A, B = np.meshgrid(a, a, indexing="ij")
    cases = []
    for t in range(4):
        c1, c2 = A.copy(), B.copy()
        if t == 0:
            c1 = np.where(A > 0, A / 2, A)
        if t == 1:
            c2 = np.where(B > 0, B / 2, B)
        r1, r2 = np.rint(c1).astype(np.int64), np.rint(c2).astype(np.int64)          # (halves round to even, like round(c, 0))
        comb = lg[r1 + r2] - lg[r1] - lg[r2]
        near = np.zeros(A.shape, dtype=bool)
        for lik in (c1 * l_ok + c2 * l_err, (c1 + c2) * l_half, c2 * l_ok + c1 * l_err):
            v = -10 * (lik + comb)
            near |= (np.abs(v - np.rint(v)) < 4e-7) & (comb != 0)
        ii, jj = np.nonzero(near)
        cases += [(t, int(x), int(y)) for x, y in zip(ii, jj)]
    rng = np.random.default_rng(11)
    for _ in range(200):
        cases.append((int(rng.integers(4)), int(rng.integers(0, 100001)), int(rng.integers(0, 100001))))
    for _ in range(40):
        cases.append((int(rng.integers(4)), int(rng.integers(0, 1000001)), int(rng.integers(0, 1000001))))
    gtc = {"0/0": 0, "0/1": 1, "1/1": 2, "./.": 3}
    rows = []
    for t, x, y in cases:
        gt, pl = ref_geno.likelihood([x, y], types[t], 3, e)
        rows.append((t, x, y, 3, gtc[gt], int(pl[0]), int(pl[1]), int(pl[2])))
    np.savez_compressed(f"{out}/lik_boundary.npz", cases=np.array(rows, dtype=np.int64), err=np.full(len(rows), e))
    print(f"lik_boundary: {len(rows)} known answers ({len(rows) - 240} next to an integer boundary)")


# ----------------------------------------------------------------------------------------------
# G5 VCF parsing
# ----------------------------------------------------------------------------------------------

def make_vcf():
    out = f"{HERE}/vcf"
    os.makedirs(out, exist_ok=True)
    hdr = ["##fileformat=VCFv4.2\n", "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"old\">\n",
           "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"t\">\n",
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\n"]
    seq60 = "ACGT" * 15
    rows = [
        "1\t1000\ta\tN\t<DEL>\t.\t.\tSVTYPE=DEL;END=1500;SVLEN=-500\tGT\t0/1\t1/1\n",       # SVTYPE first
        "1\t2000\tb\tN\t<DEL>\t.\t.\tEND=2600;SVTYPE=DEL\n",                                 # SVTYPE last, 8 cols
        "1\t3000\tc\tN\t<DEL>\t.\t.\tSVLEN=-40;SVTYPE=DEL;END=3040\tGT\t0/1\n",             # len < 50 -> ./.
        "1\t4000\td\tN\t<DEL>\t.\t.\tX=1;END=4100;SVTYPE=DEL;Y=2\tGT\t0/1\n",               # both in the middle
        f"1\t5000\te\tN\t{seq60}\t.\t.\tSVTYPE=INS;END=5001\tGT\t0/1\n",
        f"2\t5000\tf\tN\t{seq60}A\t.\t.\tSVTYPE=INS\n",                                      # same POS other chrom (Q8)
        f"1\t5000\tg\tN\t{seq60}C\t.\t.\tSVTYPE=INS;END=5001\n",                            # third INS at POS 5000
        "1\t6000\th\tN\t<INS>\t.\t.\tSVTYPE=INS;SEQ=" + seq60 + "\n",                        # symbolic: len 5 (Q12)
        "1\t7000\ti\tN\t<INV>\t.\t.\tSVTYPE=INV;END=7800\tGT\t1/1\n",
        "1\t8000\tj\tN\tN[2:9000[\t.\t.\tSVTYPE=BND\n",
        "1\t8100\tk\tN\tN]2:9100]\t.\t.\tSVTYPE=BND\n",
        "1\t8200\tl\tN\t[2:9200[N\t.\t.\tSVTYPE=BND\n",
        "1\t8300\tm\tN\t]2:9300]N\t.\t.\tSVTYPE=BND\n",
        "1\t8400\tn\tN\t<BND>\t.\t.\tSVTYPE=BND\n",                                          # wrong_format
        "1\t9000\to\tN\t<DUP>\t.\t.\tSVTYPE=DUP;END=9900\n",                                 # unsupported
        "1\t9500\tp\tN\t<DEL>\t.\t.\tEND=9900\n",                                            # no SVTYPE
        "1\t9700\tq\tN\t<DEL>\t.\t.\tSVTYPE=DEL;END=9990\tGT\t0/1\n",                        # no informative aln
    ]
    D = {
        "1:DEL-1000-1500": [["x\n"] * 9, ["y\n"] * 4],
        "1:DEL-2000-2600": [[], ["y\n"] * 7],
        "1:DEL-3000-3040": [["x\n"] * 9, ["y\n"] * 4],
        "1:DEL-4000-4100": [["x\n"] * 5, []],
        "1:INS-5000-1": [["x\n"] * 6, ["y\n"] * 6],
        "2:INS-5000-2": [["x\n"] * 1, ["y\n"] * 1],
        "1:INS-5000-3": [[], ["y\n"] * 3],
        "1:INS-6000-1": [["x\n"] * 6, ["y\n"] * 6],
        "1:INV-7000-7800": [["x\n"] * 3, ["y\n"] * 3],
        "1:BND-8000[2:9000[": [["x\n"] * 2, ["y\n"] * 12],
        "1:BND-8100]2:9100]": [["x\n"] * 12, ["y\n"] * 2],
        "1:BND-[2:9200[8200": [["x\n"] * 10, ["y\n"] * 10],
        "1:BND-]2:9300]8300": [["x\n"] * 1, ["y\n"] * 1],
        "2:BND-9000[1:8000[": [["x\n"] * 5, []],          # orphan key, matches no row
        "wrong_format": [["x\n"] * 5, ["y\n"] * 5],
        "unsupported_type": [["x\n"] * 5, ["y\n"] * 5],
    }
    with open(f"{out}/cases.vcf", "w") as fh:
        fh.write("".join(hdr + rows))
    with open(f"{out}/cases_informative_aln.json", "w") as fh:
        fh.write(json.dumps(D, sort_keys=True, indent=4))
    for ms, e, tag in ((3, None, "ms3"), (1, None, "ms1"), (3, 0.001, "ms3_e1e-3"), (0, None, "ms0")):
        rc, so = run_ref_genotype(f"{out}/cases_informative_aln.json", f"{out}/cases.vcf", f"{out}/ref_{tag}.vcf", ms, e)
        assert rc == 0
        with open(f"{out}/ref_{tag}.stdout", "w") as fh:
            fh.write(so)
    # crash case: unsupported type without END -> IndexError in the reference
    with open(f"{out}/err_no_end.vcf", "w") as fh:
        fh.write("".join(hdr) + "1\t9000\to\tN\t<DUP>\t.\t.\tSVTYPE=DUP\n")
    rc, so = run_ref_genotype(f"{out}/cases_informative_aln.json", f"{out}/err_no_end.vcf", f"{out}/_tmp.vcf")
    assert rc == 1
    os.remove(f"{out}/_tmp.vcf")
    print("vcf: ok")


def make_vcffuzz():
    """golden/vcffuzz/cases.json: small VCFs whose rows are mutations of ordinary SV rows (INFO fields shuffled, dropped, doubled;
    END / POS / ALT / SVTYPE edited; 8, 9 and 11 columns; blank columns) and, for each, a dictionary of informative alignments that
    names the keys the untouched rows would get — through the reference's predict-genotype.py: its output text, stdout, or the
    class of the exception it died with."""
    import random
    out = f"{HERE}/vcffuzz"
    os.makedirs(out, exist_ok=True)
    rng = random.Random(20260517)
    hdr = ["##fileformat=VCFv4.2\n", "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"old\">\n", "##contig=<ID=1>\n",
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"]
    seq = "ACGTTGCA" * 9

    def base_row():
        c = rng.choice(("1", "2", "chrX"))
        pos = rng.choice((1000, 1000, 2000, 5000, 123456))
        t = rng.choice(("DEL", "DEL", "INS", "INS", "INV", "BND", "DUP"))
        if t == "DEL":
            end = pos + rng.choice((30, 49, 50, 51, 700))
            return [c, str(pos), "id", "N", "<DEL>", ".", "PASS", f"SVTYPE=DEL;END={end};SVLEN={pos - end}"], f"{c}:DEL-{pos}-{end}"
        if t == "INV":
            end = pos + rng.choice((10, 50, 900))
            return [c, str(pos), "id", "N", "<INV>", ".", "PASS", f"SVTYPE=INV;END={end}"], f"{c}:INV-{pos}-{end}"
        if t == "INS":
            alt = rng.choice((seq, seq[:49], seq[:50], "<INS>", seq + "ACGT"))
            return [c, str(pos), "id", "N", alt, ".", "PASS", f"SVTYPE=INS;END={pos};SVLEN={len(alt)}"], f"{c}:INS-{pos}-"
        if t == "BND":
            c2, p2 = rng.choice(("1", "3")), rng.choice((77, 9000))
            alt = rng.choice((f"N[{c2}:{p2}[", f"N]{c2}:{p2}]", f"[{c2}:{p2}[N", f"]{c2}:{p2}]N", "<BND>", f"N[{p2}["))
            key = {"N[": f"{c}:BND-{pos}[{c2}:{p2}[", "N]": f"{c}:BND-{pos}]{c2}:{p2}]", "[" + c2: f"{c}:BND-[{c2}:{p2}[{pos}", "]" + c2: f"{c}:BND-]{c2}:{p2}]{pos}"}.get(alt[:2], "wrong_format")
            return [c, str(pos), "id", "N", alt, ".", "PASS", "SVTYPE=BND"], key
        return [c, str(pos), "id", "N", "<DUP>", ".", "PASS", f"SVTYPE=DUP;END={pos + 400}"], "unsupported_type"

    def mutate(cols):
        cols = list(cols)
        for _ in range(rng.choice((0, 1, 1, 2))):
            op = rng.randrange(12)
            f = cols[7].split(";")
            if op == 0:
                rng.shuffle(f); cols[7] = ";".join(f)
            elif op == 1 and len(f) > 1:
                del f[rng.randrange(len(f))]; cols[7] = ";".join(f)
            elif op == 2:
                f.insert(rng.randrange(len(f) + 1), rng.choice(("X=1", "SVTYPE=DEL", "END=1", "MATEID=a;b", "SVTYPE", "XEND=5", "END=abc", "CIEND=0,1"))); cols[7] = ";".join(f)
            elif op == 3:
                cols[7] = cols[7].replace("END=", rng.choice(("END=-", "end=", "END= ", "END=0")), 1)
            elif op == 4:
                cols[1] = rng.choice(("0", "x", "", "1e3", " 7", "007"))
            elif op == 5:
                cols[4] = rng.choice(("<DEL>", "N", "", ".", "A[", "[[", "]1:5]", "N[1:5", "[1:5[N]", "<INS>" * 12))
            elif op == 6:
                cols = cols + rng.choice((["GT", "0/1"], ["GT:DP", "0/1:3", "1/1:4"], ["GT"], []))
            elif op == 7 and len(cols) > 8:
                cols = cols[:8]
            elif op == 8:
                cols[7] = rng.choice(("", ".", "SVTYPE=", ";", "SVTYPE=DEL", "END=5;SVTYPE=INV;", "SVTYPE=INS;SVTYPE=DEL;END=9"))
            elif op == 9:
                cols[0] = rng.choice(("", "1", "chr 1", "1:2"))
            elif op == 10:
                cols[rng.randrange(len(cols))] = ""
            else:
                cols[7] = cols[7] + rng.choice((";", ";;", ";END=77", ";SVTYPE=BND"))
        return cols

    cases = []
    tdir = tempfile.mkdtemp()
    for i in range(260):
        rows, D = [], {}
        ins_n = {}
        for _ in range(rng.choice((1, 2, 3, 5))):
            cols, key = base_row()
            if key.endswith("-") and key.count(":INS-"):
                ins_n[cols[1]] = ins_n.get(cols[1], 0) + 1
                key += str(ins_n[cols[1]])
            if rng.random() < 0.8:
                D[key] = [["x\n"] * rng.choice((0, 1, 3, 9, 40)), ["y\n"] * rng.choice((0, 2, 7, 40))]
            if rng.random() < 0.6:
                cols = mutate(cols)
            rows.append("\t".join(cols) + "\n")
        if rng.random() < 0.1:
            rows[-1] = rows[-1].rstrip("\n")                       # no final newline
        vcf = "".join(hdr + rows)
        with open(f"{tdir}/c.vcf", "w") as fh:
            fh.write(vcf)
        with open(f"{tdir}/c.json", "w") as fh:
            fh.write(json.dumps(D, sort_keys=True, indent=4))
        ms = rng.choice((3, 3, 1, 0))
        if os.path.exists(f"{tdir}/o.vcf"):
            os.remove(f"{tdir}/o.vcf")
        cmd = [sys.executable, f"{REF}/predict-genotype.py", "-d", f"{tdir}/c.json", "-v", f"{tdir}/c.vcf", "-o", f"{tdir}/o.vcf", "--minsupport", str(ms)]
        p = subprocess.run(cmd, capture_output=True, text=True)
        case = {"vcf": vcf, "counts": {k: [len(v[0]), len(v[1])] for k, v in D.items()}, "minsupport": ms, "rc": p.returncode}
        if p.returncode == 0:
            case["out"] = open(f"{tdir}/o.vcf").read()
            case["stdout"] = p.stdout
        else:
            assert p.returncode == 1
            case["error"] = p.stderr.strip().splitlines()[-1].split(":")[0]
        cases.append(case)
    with open(f"{out}/cases.json", "w") as fh:
        json.dump(cases, fh, indent=0)
    from collections import Counter
    print("vcffuzz:", len(cases), "cases,", Counter(c.get("error", "ok") for c in cases))


def make_graphfuzz():
    """golden/graphfuzz/cases.json: tests/graph_fuzz.py's random graphs and walks (seeds 9000..9039, 500 lines each) through the
    reference's filter-alignments.py — per case the sha256 of the inputs (the generator is deterministic; the tests regenerate them),
    the per-SV list lengths and the sha256 of the JSON the reference wrote."""
    import hashlib
    sys.path.insert(0, os.path.dirname(HERE))
    import graph_fuzz
    out = f"{HERE}/graphfuzz"
    os.makedirs(out, exist_ok=True)
    tdir = tempfile.mkdtemp()
    cases = []
    for seed in range(9000, 9040):
        edges, alt, lines = graph_fuzz.make_case(seed, 500)
        with open(f"{tdir}/g_svs_edges.json", "w") as fh:
            fh.write(json.dumps(edges, indent=4))
        with open(f"{tdir}/g.gfa", "w") as fh:
            fh.write("H\tVN:Z:1.0\n")
            for n in sorted({x for k in edges for x in (k.split("@")[0], k.split("@")[2])} | set(alt)):
                fh.write(f"S\t{n}\t{'ACGT' * (alt[n] // 4) + 'A' * (alt[n] % 4) if n in alt else '*'}\n")
        with open(f"{tdir}/g.gaf", "w") as fh:
            fh.write("".join(lines))
        js = f"{tdir}/g_informative_aln.json"
        if os.path.exists(js):
            os.remove(js)
        rc, err = run_ref_filter(f"{tdir}/g.gaf", f"{tdir}/g.gfa", f"{tdir}/g")
        assert rc == 0, err
        text = open(js).read()
        d = json.loads(text)
        h_in = hashlib.sha256((json.dumps(edges, sort_keys=True) + json.dumps(alt, sort_keys=True) + "".join(lines)).encode()).hexdigest()
        cases.append({"seed": seed, "n_lines": 500, "inputs_sha256": h_in, "json_sha256": hashlib.sha256(text.encode()).hexdigest(),
                      "counts": {k: [len(v[0]), len(v[1])] for k, v in d.items()}})
    with open(f"{out}/cases.json", "w") as fh:
        json.dump(cases, fh, indent=0, sort_keys=True)
    print("graphfuzz:", len(cases), "cases,", sum(sum(map(sum, c["counts"].values())) for c in cases), "informative alignments in all")


def _write_graph_files(tdir, edges, alt):
    with open(f"{tdir}/g_svs_edges.json", "w") as fh:
        fh.write(json.dumps(edges, indent=4))
    with open(f"{tdir}/g.gfa", "w") as fh:
        fh.write("H\tVN:Z:1.0\n")
        for n in sorted({x for k in edges for x in (k.split("@")[0], k.split("@")[2])} | set(alt)):
            fh.write(f"S\t{n}\t{'ACGT' * (alt[n] // 4) + 'A' * (alt[n] % 4) if n in alt else '*'}\n")


def make_longpath():
    """golden/longpath/cases.json: tests/longpath_fuzz.py's graphs (>= 2 000 nodes) and long walks (65..216 nodes, clean for 64 nodes, ONE
    late event) through the reference's filter-alignments.py — seeds 7000..7003, 10 long lines each (40 in all, with the short lines the
    generator puts between them): per case the sha256 of the inputs, the per-SV list lengths and the sha256 of the JSON the reference
    wrote; and for every `fatal` line of the generator (an insertion node the GFA lacks) the class of the exception it died with."""
    import hashlib
    sys.path.insert(0, os.path.dirname(HERE))
    import longpath_fuzz
    out = f"{HERE}/longpath"
    os.makedirs(out, exist_ok=True)
    tdir = tempfile.mkdtemp()
    cases = []
    for seed in range(7000, 7004):
        edges, alt, lines, fatal = longpath_fuzz.make_case(seed, 10, 3)
        _write_graph_files(tdir, edges, alt)
        with open(f"{tdir}/g.gaf", "w") as fh:
            fh.write("".join(lines))
        js = f"{tdir}/g_informative_aln.json"
        if os.path.exists(js):
            os.remove(js)
        rc, err = run_ref_filter(f"{tdir}/g.gaf", f"{tdir}/g.gfa", f"{tdir}/g")
        assert rc == 0, err
        text = open(js).read()
        d = json.loads(text)
        errs = []
        for f in fatal:
            with open(f"{tdir}/f.gaf", "w") as fh:
                fh.write("".join(lines[:4] + [f] + lines[4:6]))
            rc, err = run_ref_filter(f"{tdir}/f.gaf", f"{tdir}/g.gfa", f"{tdir}/g")
            assert rc == 1, (rc, err)
            errs.append(err.split(":")[0])
        h_in = hashlib.sha256((json.dumps(edges, sort_keys=True) + json.dumps(alt, sort_keys=True) + "".join(lines) + "".join(fatal)).encode()).hexdigest()
        cases.append({"seed": seed, "n_long": 10, "n_fatal": 3, "n_lines": len(lines), "inputs_sha256": h_in, "json_sha256": hashlib.sha256(text.encode()).hexdigest(),
                      "counts": {k: [len(v[0]), len(v[1])] for k, v in d.items()}, "fatal_errors": errs,
                      "events": [l.split("\t", 1)[0] for l in lines if "_k" in l.split("\t", 1)[0]]})
    with open(f"{out}/cases.json", "w") as fh:
        json.dump(cases, fh, indent=0, sort_keys=True)
    print("longpath:", len(cases), "cases,", sum(len(c["events"]) for c in cases), "long lines,", sum(sum(map(sum, c["counts"].values())) for c in cases), "informative alignments in all;",
          "fatal:", sorted({e for c in cases for e in c["fatal_errors"]}))


def make_longtail():
    """golden/longtail/cases.json: tests/longpath_fuzz.py: make_tail_case — lines longer than the main kernel's 8 KB stage with ONE event in
    a 6..40 KB tail at a boundary position (a carriage return, "d:", an id:f: tag, bytes >= 0x80, the terminator, none) — on a synthetic
    graph (tools/synth.py, 800 alignments x 200 SVs), seeds 0..3 with 120 mutated lines each, through the reference's filter-alignments.py:
    sha256 of the inputs, per-SV list lengths, sha256 of its JSON, and the exception class for each of the case's fatal texts."""
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, f"{ROOT}/tools")
    import longpath_fuzz
    import synth
    out = f"{HERE}/longtail"
    os.makedirs(out, exist_ok=True)
    cases = []
    for seed in range(4):
        tmp = tempfile.mkdtemp()
        pre = f"{tmp}/s"
        inf = synth.generate(pre, 800, 200, 2, "mixed", 700 + seed, write_gaf=False, return_gaf=True)
        base = inf["gaf"].tobytes().split(b"\n")[:-1]
        text, fatal = longpath_fuzz.make_tail_case(300 + seed, base, 120)
        with open(pre + ".gaf", "wb") as fh:
            fh.write(text)
        rc, err = run_ref_filter(pre + ".gaf", pre + ".gfa", pre)
        assert rc == 0, err
        js = open(pre + "_informative_aln.json").read()
        d = json.loads(js)
        errs = []
        for f in fatal[:5]:
            with open(f"{tmp}/f.gaf", "wb") as fh:
                fh.write(f)
            rc, err = run_ref_filter(f"{tmp}/f.gaf", pre + ".gfa", pre)
            assert rc == 1, (rc, err)
            errs.append(err.split(":")[0])
        cases.append({"seed": seed, "synth": [800, 200, 2, "mixed", 700 + seed], "tail_seed": 300 + seed, "n_mut": 120,
                      "inputs_sha256": hashlib.sha256(text + b"".join(fatal[:5])).hexdigest(), "json_sha256": hashlib.sha256(js.encode()).hexdigest(),
                      "counts": {k: [len(v[0]), len(v[1])] for k, v in d.items()}, "fatal_errors": errs, "gaf_bytes": len(text)})
        shutil.rmtree(tmp)
    with open(f"{out}/cases.json", "w") as fh:
        json.dump(cases, fh, indent=0, sort_keys=True)
    print("longtail:", len(cases), "cases,", sum(c["gaf_bytes"] for c in cases), "bytes of GAF,", sum(sum(map(sum, c["counts"].values())) for c in cases), "informative alignments;",
          "fatal:", sorted({e for c in cases for e in c["fatal_errors"]}))


def ref_filter_inproc(gaf, gfa, prefix):
    """The reference's filter-alignments.py main() in THIS process (the module is imported above; main() parses sys.argv itself,
    filter-alignments.py:74) -> ("ok", per-SV list lengths of the JSON it wrote) | ("died", exception class).  For groups of thousands
    of tiny cases, where a process per case is most of the time."""
    js = prefix + "_informative_aln.json"
    if os.path.exists(js):
        os.remove(js)
    argv = sys.argv
    sys.argv = ["filter-alignments.py", "-a", gaf, "-g", gfa, "-p", prefix]
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ref_filter.main(None)
    except Exception as e:
        return ("died", type(e).__name__)
    finally:
        sys.argv = argv
    d = json.load(open(js))
    return ("ok", {k: [len(v[0]), len(v[1])] for k, v in d.items()})


def _pack_cases(frags, verdicts):
    """fragments (bytes) + the reference's verdicts -> JSON-able dict: the texts as one zlib blob (mutants of 40 lines deflate to a few
    bytes each) so that the fixture does not depend on the mutator's random stream staying the same"""
    import base64
    import zlib
    return {"lengths": [len(f) for f in frags], "text_zlib_b64": base64.b64encode(zlib.compress(b"".join(frags), 9)).decode(),
            "text_sha256": hashlib.sha256(b"".join(frags)).hexdigest(),
            "verdicts": [v[1] if v[0] == "ok" else v[1] for v in verdicts]}          # a dict = accepted (its counts), a string = the exception class


def _testdir_tmp():
    tmp = tempfile.mkdtemp()
    for ext in (".gfa", "_svs_edges.json"):
        os.symlink(f"{HERE}/testdir/test{ext}", os.path.join(tmp, "test" + ext))
    return tmp


def make_blanks():
    """golden/blanks/blanks.json (r06): which bytes int() / float() / str.rstrip() of the reference take for blanks — each of 0x09-0x0D,
    0x1C-0x1F, 0x20, 0x00, 0x7F in front of and behind each of the nine decimal columns, at both ends of an id:f: value (last tag / a tag
    in front of others), at the line's end, for three lines of testdir/test.gaf (tests/alphabet_fuzz.py: blank_cases); and the same bytes
    where the main kernel's 8 KB stage ends before them: behind column 8 of a line whose read name is 8 300 bytes long, at the end of an
    id:f: value and at the line's end behind a cg:Z: tag of 9 000 bytes.  Every case through the reference's filter-alignments.py."""
    sys.path.insert(0, os.path.dirname(HERE))
    import alphabet_fuzz as AF
    base = open(f"{HERE}/testdir/test.gaf", "rb").read().splitlines(keepends=True)
    cases = [(f"l{i}_{lab}", b) for i in (0, 3, 4) for lab, b in AF.blank_cases(base[i])]
    for ch in (0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x1C, 0x1D, 0x1E, 0x1F, 0x20, 0x00, 0x7F):
        c1 = bytes([ch])
        cols = base[0].rstrip(b"\n").split(b"\t")
        x = list(cols); x[0] = b"r" * 8300; x[8] += c1
        cases.append((f"past_stage_col8_back_{ch:02x}", b"\t".join(x) + b"\n"))
        x = list(cols); x[0] = b"r" * 8300; x[7] = c1 + x[7]
        cases.append((f"past_stage_col7_front_{ch:02x}", b"\t".join(x) + b"\n"))
        cases.append((f"past_stage_idf_back_{ch:02x}", b"\t".join(cols + [b"cg:Z:" + b"12M3D" * 1800, b"id:f:0.9" + c1]) + b"\n"))
        cases.append((f"past_stage_idf_front_{ch:02x}", b"\t".join(cols + [b"cg:Z:" + b"12M3D" * 1800, b"id:f:" + c1 + b"0.9", b"zz:i:1"]) + b"\n"))
        cases.append((f"past_stage_end_{ch:02x}", b"\t".join(cols + [b"cg:Z:" + b"12M3D" * 1800]) + c1 + b"\n"))
    tmp = _testdir_tmp()
    verdicts = []
    for _, frag in cases:
        open(f"{tmp}/f.gaf", "wb").write(frag)
        verdicts.append(ref_filter_inproc(f"{tmp}/f.gaf", f"{tmp}/test.gfa", f"{tmp}/test"))
    out = _pack_cases([f for _, f in cases], verdicts)
    out["labels"] = [lab for lab, _ in cases]
    out["graph"] = "testdir/test.gfa + testdir/test_svs_edges.json"
    out["python"] = sys.version.split()[0]
    os.makedirs(f"{HERE}/blanks", exist_ok=True)
    json.dump(out, open(f"{HERE}/blanks/blanks.json", "w"), indent=0, sort_keys=True)
    died = {}
    for (lab, _), v in zip(cases, verdicts):
        if v[0] == "died":
            died[lab.rsplit("_", 1)[1]] = died.get(lab.rsplit("_", 1)[1], 0) + 1
    print("blanks:", len(cases), "cases; died per byte:", died)
    shutil.rmtree(tmp)


def make_fuzz7(n_cases=12000, seed=20261005):
    """golden/fuzz7/fuzz7.json (r06): mutants of testdir/test.gaf over the FULL 7-bit alphabet (tests/alphabet_fuzz.py: mutate7 — any byte
    anywhere, blank-like bytes around the decimal columns and id:f: values, numbers of 19..4301 digits, node names that are arithmetic),
    each evaluated by the reference's filter-alignments.py.  The r05 fixture (golden/fuzz) drew from an 18-byte alphabet."""
    sys.path.insert(0, os.path.dirname(HERE))
    import alphabet_fuzz as AF
    base = open(f"{HERE}/testdir/test.gaf", "rb").read().splitlines(keepends=True)
    frags = AF.mutants(base, n_cases, seed)
    tmp = _testdir_tmp()
    verdicts = []
    for frag in frags:
        open(f"{tmp}/f.gaf", "wb").write(frag)
        verdicts.append(ref_filter_inproc(f"{tmp}/f.gaf", f"{tmp}/test.gfa", f"{tmp}/test"))
    out = _pack_cases(frags, verdicts)
    out.update({"seed": seed, "graph": "testdir/test.gfa + testdir/test_svs_edges.json", "python": sys.version.split()[0],
                "int_max_str_digits": sys.get_int_max_str_digits() if hasattr(sys, "get_int_max_str_digits") else 0})
    os.makedirs(f"{HERE}/fuzz7", exist_ok=True)
    json.dump(out, open(f"{HERE}/fuzz7/fuzz7.json", "w"), indent=0, sort_keys=True)
    errs = {}
    for v in verdicts:
        if v[0] == "died":
            errs[v[1]] = errs.get(v[1], 0) + 1
    print(f"fuzz7: {len(frags)} cases: {sum(v[0] == 'ok' for v in verdicts)} accepted ({sum(v[0] == 'ok' and bool(v[1]) for v in verdicts)} with hits), died: {errs}")
    shutil.rmtree(tmp)


def make_hg002shape(n_reads=200_000):
    """golden/hg002shape/hg002shape.json (r06): BASELINE configs[4]'s SHAPE without minigraph and without the data (tools/synth.py:
    make_svs_hg002 — 12.8 k DEL / INS of 50 bp .. 10 kb on the 24 GRCh37 contigs at their real lengths, SV deserts, clusters — and
    svjg_synth_gaf_reads — 30x of ~20 kb reads walked from genome positions, nine lines in ten single-node paths).
      1. The GRAPH is built by the reference's construct-graph.py from the generated VCF and a FASTA stand-in of the contigs' lengths
         (3.1 GB of 'N': the filter never reads reference sequences); its _svs_edges.json must equal tools/synth.py's byte for byte and
         its GFA after eliding the reference sequences — the fixture holds the sha256 of both, so the GPU box can regenerate the graph
         and know it is the one the reference built.
      2. The first n_reads lines of the GAF through the reference's filter-alignments.py and predict-genotype.py: per-SV counts,
         sha256 of the JSON and of the VCF, the reference's run times."""
    import time
    sys.path.insert(0, f"{ROOT}/tools")
    import synth
    out = f"{HERE}/hg002shape"
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp(dir="/tmp")
    pre = f"{tmp}/hg"
    inf = synth.generate_hg002(pre, n_reads=n_reads)
    t0 = time.time()
    with open(f"{tmp}/ref.fa", "w") as fh:
        for c, l in zip(inf["chroms"], inf["chrom_len"]):
            fh.write(f">{c}\n")
            fh.write("N" * l)
            fh.write("\n")
    os.makedirs(f"{tmp}/r")
    t1 = time.time()
    subprocess.run([sys.executable, f"{REF}/construct-graph.py", "-v", pre + ".vcf", "-r", f"{tmp}/ref.fa", "-o", f"{tmp}/r/hg.gfa"], check=True, capture_output=True)
    t_construct = time.time() - t1
    os.remove(f"{tmp}/ref.fa")
    assert open(pre + "_svs_edges.json").read() == open(f"{tmp}/r/hg_svs_edges.json").read(), "edges JSON differs from construct-graph.py's"
    h_ref, h_own = hashlib.sha256(), hashlib.sha256()
    with open(f"{tmp}/r/hg.gfa") as fr, open(pre + ".gfa") as fo:
        for lr in fr:
            if lr.startswith("S"):
                c = lr.rstrip("\n").split("\t", 2)
                if "." not in c[1].split(":")[-1]:
                    lr = f"S\t{c[1]}\t*\n"
            lo = fo.readline()
            assert lr == lo, (lr[:200], lo[:200])
            h_ref.update(lr.encode()); h_own.update(lo.encode())
        assert fo.readline() == ""
    os.remove(f"{tmp}/r/hg.gfa")
    # the filter and the genotyper of the reference on the sample (the graph files are now known to be the reference's own)
    t1 = time.time()
    rc, err = run_ref_filter(pre + ".gaf", pre + ".gfa", pre)
    assert rc == 0, err
    t_filter = time.time() - t1
    js = open(pre + "_informative_aln.json").read()
    d = json.loads(js)
    t1 = time.time()
    rc, so = run_ref_genotype(pre + "_informative_aln.json", pre + ".vcf", pre + "_genotype.vcf")
    assert rc == 0
    t_geno = time.time() - t1
    vcf = open(pre + "_genotype.vcf", "rb").read()
    gaf = open(pre + ".gaf", "rb").read()
    lines = gaf.split(b"\n")[:-1]
    k = [ln.split(b"\t")[5].count(b">") + ln.split(b"\t")[5].count(b"<") for ln in lines]
    fix = {"seed": synth.HG002_SEED, "n_reads": n_reads, "n_sv": inf["n_sv"], "n_nodes": inf["n_nodes"], "n_edge_keys": inf["n_edge_keys"],
           "edges_json_sha256": hashlib.sha256(open(pre + "_svs_edges.json", "rb").read()).hexdigest(),
           "gfa_elided_sha256": h_own.hexdigest(), "vcf_in_sha256": hashlib.sha256(open(pre + ".vcf", "rb").read()).hexdigest(),
           "gaf_sha256": hashlib.sha256(gaf).hexdigest(), "gaf_bytes": len(gaf), "single_node_lines": int(sum(x == 1 for x in k)),
           "json_sha256": hashlib.sha256(js.encode()).hexdigest(), "json_bytes": len(js), "vcf_sha256": hashlib.sha256(vcf).hexdigest(),
           "genotyped": so.strip(), "counts": {key: [len(v[0]), len(v[1])] for key, v in d.items()},
           "reference_seconds": {"construct_graph": round(t_construct, 1), "filter": round(t_filter, 2), "genotype": round(t_geno, 2)},
           "graph_built_by": "construct-graph.py of the reference, from the generated VCF + a FASTA of N at GRCh37 lengths"}
    assert h_ref.hexdigest() == h_own.hexdigest()
    # ... and the WHOLE block of bench.py (30x: synth.HG002_READS lines, 750 MB) through the reference's filter and genotyper: the reference can run
    # this shape at full size (nine lines in ten are skipped at filter-alignments.py:133-134; ~620 k informative alignments)
    full_n = synth.HG002_READS
    gaf_full = synth.gaf_bytes(inf["tables"], synth.HG002_SEED, 0, full_n, threads=8, shape="reads")
    assert hashlib.sha256(gaf_full[: len(gaf)].tobytes()).hexdigest() == fix["gaf_sha256"]          # (the sample is the stream's head)
    gaf_full.tofile(pre + ".gaf")
    t1 = time.time()
    rc, err = run_ref_filter(pre + ".gaf", pre + ".gfa", pre)
    assert rc == 0, err
    t_filter_full = time.time() - t1
    js = open(pre + "_informative_aln.json").read()
    d = json.loads(js)
    t1 = time.time()
    rc, so = run_ref_genotype(pre + "_informative_aln.json", pre + ".vcf", pre + "_genotype.vcf")
    assert rc == 0
    t_geno_full = time.time() - t1
    fix["full"] = {"n_reads": full_n, "gaf_bytes": int(gaf_full.size), "gaf_sha256": hashlib.sha256(gaf_full.tobytes()).hexdigest(),
                   "json_sha256": hashlib.sha256(js.encode()).hexdigest(), "json_bytes": len(js),
                   "vcf_sha256": hashlib.sha256(open(pre + "_genotype.vcf", "rb").read()).hexdigest(), "genotyped": so.strip(),
                   "counts": {key: [len(v[0]), len(v[1])] for key, v in d.items()},
                   "reference_seconds": {"filter": round(t_filter_full, 1), "genotype": round(t_geno_full, 1),
                                         "where": "build container, 1 core of an Intel Xeon @ 2.1 GHz, Python " + sys.version.split()[0]}}
    del d, js, gaf_full
    json.dump(fix, open(f"{out}/hg002shape.json", "w"), indent=0, sort_keys=True)
    print("hg002shape:", {k2: v for k2, v in fix.items() if k2 not in ("counts", "full")}, len(fix["counts"]), "SVs with informative alignments")
    print("hg002shape, whole block:", {k2: v for k2, v in fix["full"].items() if k2 != "counts"}, len(fix["full"]["counts"]), "SVs with informative alignments")
    shutil.rmtree(tmp)


def make_contigs():
    """golden/contigs: contig names shaped like the GRCh38 analysis set's — HLA-DRB1*15:03:01:01 and HLA-A*01:01:01:01 (':', '*' and '-'
    INSIDE the contig part: the reference takes the LAST ':' field of a node name, filter-alignments.py:328-349), chrUn_JTFH01001998v1_decoy,
    chr6_GL000250v2_alt, chrEBV, chr6 — on two small graphs, `hla` (all six contigs) and `ucsc` (the four without a colon), with walks in
    both directions over reference and insertion nodes, links between contigs, names the graph lacks on contigs it has and on contigs it
    has not, single-node paths; the reference's JSON for both is committed."""
    import random
    out = f"{HERE}/contigs"
    os.makedirs(out, exist_ok=True)
    for tag, chroms in (("hla", ["chr6", "chr6_GL000250v2_alt", "chrUn_JTFH01001998v1_decoy", "chrEBV", "HLA-DRB1*15:03:01:01", "HLA-A*01:01:01:01"]),
                        ("ucsc", ["chr6", "chr6_GL000250v2_alt", "chrUn_JTFH01001998v1_decoy", "chrEBV"])):
        rng = random.Random(len(chroms))
        edges, alt, ref, length = {}, {}, {}, {}
        n_sv = [0]

        def add(l, sl, r, sr, ents):
            edges.setdefault("@".join((l, sl, r, sr)), []).extend(ents)

        def sv(c, kind, pos):
            n_sv[0] += 1
            return f"{c}:INS-{pos}-{n_sv[0] % 4 + 1}" if kind == "INS" else f"{c}:{kind}-{pos}-{pos + 300 + n_sv[0]}"
        for c in chroms:
            cuts = sorted(rng.sample(range(300, 40000 if c != "chr6" else 31_000_000), 9))
            starts, ends = [1] + [x + 1 for x in cuts], cuts + [cuts[-1] + 2500]
            ref[c] = [f"{c}:{a}-{b}" for a, b in zip(starts, ends)]
            for n, a, b in zip(ref[c], starts, ends):
                length[n] = b - a + 1
            for i in range(9):
                add(ref[c][i], "+", ref[c][i + 1], "+", [[sv(c, "DEL", starts[i + 1]), 0]])
                if i % 3 == 0 and i + 2 < 10:
                    add(ref[c][i], "+", ref[c][i + 2], "+", [[sv(c, "DEL", starts[i + 1]), 1]])
                if i % 3 == 1:
                    an = f"{c}:{starts[i + 1]}.1"
                    alt[an] = length[an] = 120 + 10 * i
                    s_id = sv(c, "INS", starts[i + 1])
                    add(ref[c][i], "+", an, "+", [[s_id, 1]])
                    add(an, "+", ref[c][i + 1], "+", [[s_id, 1]])
                if i % 4 == 2:
                    add(ref[c][i], "+", ref[c][i + 1], "-", [[sv(c, "INV", starts[i + 1]), 1]])
        for a, b in zip(chroms, chroms[1:] + chroms[:1]):                 # breakends between the contigs
            add(ref[a][4], "+", ref[b][6], "+", [[f"{a}:BND-{ref[a][4].rsplit('-', 1)[1]}[{b}:{ref[b][6].rsplit(':', 1)[1].split('-')[0]}[", 1]])
        lines = []
        order = {}
        for c in chroms:
            seq = []
            for i, n in enumerate(ref[c]):
                pos = n.rsplit(":", 1)[1].split("-")[0]
                if f"{c}:{pos}.1" in alt:
                    seq.append(f"{c}:{pos}.1")
                seq.append(n)
            order[c] = seq
        absent = ["HLA-B*07:02:01:1-3000", "HLA-C*04:01:01:01:100-900", "chr6_GL000251v2_alt:5-4000", "chrUn_KN707606v1_decoy:1-2200", "HLA-DRB1*15:03:01:02:1-500"]

        def nlen(n):
            if n in length:
                return length[n]
            a, b = n.rsplit(":", 1)[1].split("-")
            return int(b) - int(a) + 1
        for i in range(90):
            c = chroms[i % len(chroms)]
            seq = order[c]
            k = rng.choice((1, 2, 3, 4, 5, 8, len(seq)))
            a0 = rng.randrange(0, len(seq) - k + 1)
            w = seq[a0:a0 + k]
            if i % 7 == 3 and k >= 3:                                 # over a deletion's link: leave a reference node out
                w = [x for j, x in enumerate(w) if j != 1 or "." in x] or w
            if i % 9 == 4:                                            # on to the next contig over the breakend
                c2 = chroms[(chroms.index(c) + 1) % len(chroms)]
                w = order[c][: order[c].index(ref[c][4]) + 1][-3:] + order[c2][order[c2].index(ref[c2][6]):][:3]
            if i % 11 == 5:
                w = w + [absent[i % len(absent)]]
            if i % 13 == 6:
                w = [absent[(i + 1) % len(absent)]] + w
            if i % 17 == 7:
                w = [absent[i % len(absent)]]                         # a single-node path on a contig the graph does not have
            rev = i % 2 == 1
            tl = sum(nlen(n) for n in w)
            p = "".join(("<" if rev else ">") + n for n in (reversed(w) if rev else w))
            ts, back = rng.choice((0, 5, 99, 100, 101, 250)), rng.choice((0, 7, 99, 100, 101, 250))
            te = max(tl - back, ts + 1)
            tags = "tp:A:P\tcm:i:9\ts1:i:77\ts2:i:0\tdv:f:0.0312" + ("\tcg:Z:30M2I40M" if i % 5 == 0 else "")
            lines.append(f"m64011_190830_220126/{i}/ccs\t{tl + 30}\t2\t{tl + 10}\t{'+-'[i % 2]}\t{p}\t{tl}\t{ts}\t{te}\t{max(te - ts - 9, 1)}\t{max(te - ts, 1)}\t{60 - i % 3}\t{tags}\n")
        pre = f"{out}/{tag}"
        with open(pre + "_svs_edges.json", "w") as fh:
            fh.write(json.dumps(edges, sort_keys=True, indent=4))
        with open(pre + ".gfa", "w") as fh:
            fh.write("H\tVN:Z:1.0\n")
            for n in sorted({x for k in edges for x in (k.split("@")[0], k.split("@")[2])} | set(alt)):
                fh.write(f"S\t{n}\t{'ACGT' * (alt[n] // 4) + 'A' * (alt[n] % 4) if n in alt else '*'}\n")
        with open(pre + ".gaf", "w") as fh:
            fh.write("".join(lines))
        js = pre + "_informative_aln.json"
        if os.path.exists(js):
            os.remove(js)
        rc, err = run_ref_filter(pre + ".gaf", pre + ".gfa", pre)
        assert rc == 0, err
        os.replace(js, pre + ".ref.json")
        d = json.load(open(pre + ".ref.json"))
        print(f"contigs/{tag}:", len(lines), "lines,", len(d), "SVs with informative alignments,", sum(len(v[0]) + len(v[1]) for v in d.values()), "hits")


# ----------------------------------------------------------------------------------------------
# G6 medium synthetic (needs tools/svjg_synth built)
# ----------------------------------------------------------------------------------------------

def make_synth():
    out = f"{HERE}/synth"
    os.makedirs(out, exist_ok=True)
    sys.path.insert(0, f"{ROOT}/tools")
    import synth
    import time
    res = {}
    for tag, kw in (("g6_mixed", dict(n_aln=50000, n_sv=2000, n_chrom=4, mix="mixed", seed=20260515 + 6)),
                    ("g6_del", dict(n_aln=20000, n_sv=500, n_chrom=1, mix="del", seed=20260515 + 1))):
        tmp = tempfile.mkdtemp()
        synth.generate(prefix=f"{tmp}/s", **kw)
        t0 = time.time()
        rc, err = run_ref_filter(f"{tmp}/s.gaf", f"{tmp}/s.gfa", f"{tmp}/s")
        t1 = time.time()
        assert rc == 0, err
        rc, so = run_ref_genotype(f"{tmp}/s_informative_aln.json", f"{tmp}/s.vcf", f"{tmp}/s_genotype.vcf")
        t2 = time.time()
        assert rc == 0
        D = json.load(open(f"{tmp}/s_informative_aln.json"))
        res[tag] = {
            "args": kw,
            "sha256_inputs": {f: hashlib.sha256(open(f"{tmp}/s{f}", "rb").read()).hexdigest()
                              for f in (".gaf", ".gfa", "_svs_edges.json", ".vcf")},
            "sha256_json": hashlib.sha256(open(f"{tmp}/s_informative_aln.json", "rb").read()).hexdigest(),
            "sha256_vcf": hashlib.sha256(open(f"{tmp}/s_genotype.vcf", "rb").read()).hexdigest(),
            "stdout": so,
            "ref_seconds": {"filter": round(t1 - t0, 2), "genotype": round(t2 - t1, 2)},
            "counts": {k: [len(v[0]), len(v[1])] for k, v in sorted(D.items())},
        }
        shutil.copy(f"{tmp}/s_genotype.vcf", f"{out}/{tag}.ref_genotype.vcf")
        shutil.rmtree(tmp)
    with open(f"{out}/g6.json", "w") as fh:
        json.dump(res, fh, indent=0, sort_keys=True)
    print("synth:", {k: v["ref_seconds"] for k, v in res.items()})


# ----------------------------------------------------------------------------------------------
# G7 lines shaped like real `minigraph -x lr` output (stand-in for BASELINE configs[4], which needs minigraph and HG002 reads)
# ----------------------------------------------------------------------------------------------

def make_realshape():
    """~180 GAF lines on a 600-SV graph with UCSC-style contig names: PacBio / ONT read names, the tags minigraph writes
    (tp cm s1 s2 dv, and cg:Z: / ds:Z: strings of kilobytes as with -c / --ds), paths of 2..300 nodes in both directions,
    long reads threading dozens of SV sites, a GraphAligner-style id:f: tag.  Inputs + the reference's JSON are committed."""
    out = f"{HERE}/realshape"
    os.makedirs(out, exist_ok=True)
    sys.path.insert(0, f"{ROOT}/tools")
    import synth
    tmp = tempfile.mkdtemp()
    inf = synth.generate(f"{tmp}/r", 160, 600, 3, "mixed", 20260515 + 7, return_gaf=True)
    ren = {"chr2": "chrUn_KI270742v1", "chr3": "chr22_KI270879v1_alt"}

    def rename(t):
        for a, b in ren.items():
            t = t.replace(a + ":", b + ":").replace(a + "\t", b + "\t").replace("#" + a, "#" + b).replace("\t" + a + "\t", "\t" + b + "\t")
        return t
    for ext in (".gfa", "_svs_edges.json", ".vcf"):
        with open(f"{tmp}/r{ext}") as fi, open(f"{out}/r{ext}", "w") as fo:
            fo.write(rename(fi.read()))
    rng = np.random.default_rng(7)

    def read_name(i):
        k = i % 4
        if k == 0:
            return "m64011_190830_220126/%d/ccs" % int(rng.integers(1, 180000000))
        if k == 1:
            h = "".join("0123456789abcdef"[int(x)] for x in rng.integers(0, 16, 32))
            return f"{h[:8]}-{h[8:12]}-{h[12:16]}-{h[16:20]}-{h[20:]}"
        if k == 2:
            return "SRR%d.%d" % (int(rng.integers(9000000, 9999999)), int(rng.integers(1, 4000000)))
        return "HG002_ONT_UL_%06d_ch%d_read%d" % (i, int(rng.integers(1, 512)), int(rng.integers(1, 90000)))

    def cigar(n):
        parts, left = [], n
        while left > 0:
            m = int(min(left, rng.integers(20, 900)))
            parts.append(f"{m}M")
            left -= m
            if left > 0 and rng.random() < 0.8:
                parts.append(f"{int(rng.integers(1, 12))}{'ID'[int(rng.integers(0, 2))]}")
        return "".join(parts)

    def ds(n):
        parts, left = [], n
        while left > 0:
            m = int(min(left, rng.integers(30, 1200)))
            parts.append(f":{m}")
            left -= m
            if left > 0:
                parts.append("*" + "acgt"[int(rng.integers(0, 4))] + "acgt"[int(rng.integers(0, 4))] if rng.random() < 0.5
                             else "+-"[int(rng.integers(0, 2))] + "".join("acgt"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 30)))))
        return "".join(parts)

    lines = []
    for i, ln in enumerate(rename(inf["gaf"].tobytes().decode()).splitlines()):
        c = ln.split("\t")
        c[0] = read_name(i)
        alen = int(c[10])
        tags = c[12:]
        if i % 3 == 0:
            tags.append("cg:Z:" + cigar(alen))
        if i % 5 == 0:
            tags.append("ds:Z:" + ds(alen))
        if i % 41 == 0:
            tags.insert(2, "id:f:0.9871")                               # GraphAligner writes it; float() accepts it
        lines.append("\t".join(c[:12] + tags) + "\n")
    # long reads: many consecutive reference nodes, forward and reverse (13..300 nodes; more than 64 nodes: the exact path)
    names = {}
    for ln in open(f"{out}/r.gfa"):
        if ln.startswith("S"):
            nm = ln.split("\t")[1]
            if "." not in nm.split(":")[-1]:
                names.setdefault(nm.rsplit(":", 1)[0], []).append(nm)
    for chrom in names:
        names[chrom].sort(key=lambda n: int(n.rsplit(":", 1)[1].split("-")[0]))

    def nlen(n):
        a, b = n.rsplit(":", 1)[1].split("-")
        return int(b) - int(a) + 1
    for q, (chrom, start, k) in enumerate((("chr1", 3, 13), ("chr1", 20, 33), ("chrUn_KI270742v1", 5, 64), ("chr22_KI270879v1_alt", 9, 65),
                                           ("chr1", 40, 80), ("chrUn_KI270742v1", 2, 128), ("chr22_KI270879v1_alt", 1, 160), ("chr1", 0, 300))):
        path = names[chrom][start:start + k]
        assert len(path) == k, (chrom, len(names[chrom]))
        tlen = sum(nlen(n) for n in path)
        for rev in (False, True):
            p = "".join(("<" if rev else ">") + n for n in (reversed(path) if rev else path))
            ts, te = int(rng.integers(0, 300)), tlen - int(rng.integers(0, 300))
            al = te - ts
            tags = ["tp:A:P", f"cm:i:{al // 14}", f"s1:i:{al * 3 // 4}", "s2:i:0", "dv:f:0.0031"]
            if q % 2 == 0:
                tags.append("cg:Z:" + cigar(al))
            lines.append("\t".join([read_name(1000 + 2 * q + rev), str(al + 40), "17", str(al + 17), "+", p, str(tlen), str(ts), str(te),
                                     str(al - al // 300), str(al), "60"] + tags) + "\n")
    with open(f"{out}/r.gaf", "w") as fh:
        fh.write("".join(lines))
    rc, err = run_ref_filter(f"{out}/r.gaf", f"{out}/r.gfa", f"{out}/r")
    assert rc == 0, err
    rc, so = run_ref_genotype(f"{out}/r_informative_aln.json", f"{out}/r.vcf", f"{out}/r.ref_genotype.vcf", ms=1)
    assert rc == 0
    D = json.load(open(f"{out}/r_informative_aln.json"))
    import gzip                                                         # (a long line sits in the lists of every SV it crosses: 8 MB of very repetitive text)
    with open(f"{out}/r_informative_aln.json", "rb") as fi, open(f"{out}/r.ref.json.gz", "wb") as fo:
        with gzip.GzipFile(fileobj=fo, mode="wb", mtime=0, compresslevel=9) as gz:
            shutil.copyfileobj(fi, gz)
    os.remove(f"{out}/r_informative_aln.json")
    shutil.rmtree(tmp)
    print(f"realshape: {len(lines)} lines, {max(len(l) for l in lines)} bytes the longest, {len(D)} informative SVs, {so.strip()}")


# ----------------------------------------------------------------------------------------------
# G8 which error comes first when the GAF is not UTF-8 (the reference reads it in text mode, block by block)
# ----------------------------------------------------------------------------------------------

def make_utf8order():
    import base64
    out = f"{HERE}/utf8order"
    os.makedirs(out, exist_ok=True)
    q = f"{HERE}/quirks"
    good = open(f"{q}/alt_del.gaf", "rb").read().splitlines(True)[0]
    bad = b"x\t1\t2\n"                                                # too few columns: ValueError
    per = 8192 // len(good)
    cases = {
        "bad_line_first_block_byte_far": (bad + good * 200 + b"\xff" + good, 0),
        "byte_same_block_behind_the_line": (bad + b"\xff" + good, 0),
        "byte_in_front_of_the_line": (good * 3 + b"na\xffme" + good + bad, len(good) * 4 + 5),
        "line_straddles_into_the_byte_s_block": (good * per + bad + good * 50 + b"\xff", len(good) * per),
        "line_and_byte_in_the_second_block": (good * 100 + bad + good * 10 + b"\xff", len(good) * 100),
        "line_in_the_first_block_byte_in_the_second": (good * 10 + bad + good * 100 + b"\xff", len(good) * 10),
        "no_bad_line_only_the_byte": (good * 5 + b"r\xc3(x" + good, None),
    }
    res = {}
    tmp = tempfile.mkdtemp()
    for name, (raw, off) in cases.items():
        with open(f"{tmp}/{name}.gaf", "wb") as fh:
            fh.write(raw)
        shutil.copy(f"{q}/q_svs_edges.json", f"{tmp}/{name}_svs_edges.json")
        rc, err = run_ref_filter(f"{tmp}/{name}.gaf", f"{q}/q.gfa", f"{tmp}/{name}")
        assert rc == 1
        res[name] = {"gaf": base64.b64encode(raw).decode(), "bad_line_offset": off, "error": err.split(":")[0]}
    shutil.rmtree(tmp)
    with open(f"{out}/cases.json", "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print("utf8order:", {k: v["error"] for k, v in res.items()})


# ----------------------------------------------------------------------------------------------
# BASELINE configs at full size: the reference itself on the generated files (minutes to half an hour of one core each)
# ----------------------------------------------------------------------------------------------

def _sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for b in iter(lambda: fh.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


def _timed(cmd):
    """-> (returncode, stdout, seconds, peak RSS in MB of the child)"""
    import resource
    import time
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    t = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.time() - t
    after = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    return p.returncode, p.stdout, dt, max(before, after) // 1024


def _full_case(cfg, n_aln=None, vcf_rows=None, scratch=None):
    """tools/synth.py configuration `cfg` (optionally only its first n_aln alignments, and only the first vcf_rows data rows of its
    VCF for the genotyper, whose key test is quadratic) through the reference's two scripts."""
    sys.path.insert(0, f"{ROOT}/tools")
    import synth
    base = scratch or ("/dev/shm" if os.path.isdir("/dev/shm") else None)
    tmp = tempfile.mkdtemp(prefix=f"svjg_golden_{cfg}_", dir=base)
    pre = f"{tmp}/{cfg}"
    try:
        n, n_sv, n_chrom, mix, seed = synth.CONFIGS[cfg]
        n = n_aln or n
        synth.generate(pre, n, n_sv, n_chrom, mix, seed)
        rc_f, _, t_f, rss_f = _timed([sys.executable, f"{REF}/filter-alignments.py", "-a", pre + ".gaf", "-g", pre + ".gfa", "-p", pre])
        assert rc_f == 0
        vcf = pre + ".vcf"
        if vcf_rows:
            vcf = pre + "_head.vcf"
            k = 0
            with open(pre + ".vcf") as fi, open(vcf, "w") as fo:
                for ln in fi:
                    if not ln.startswith("#"):
                        k += 1
                        if k > vcf_rows:
                            break
                    fo.write(ln)
        rc_g, so, t_g, rss_g = _timed([sys.executable, f"{REF}/predict-genotype.py", "-d", pre + "_informative_aln.json", "-v", vcf,
                                       "-o", pre + "_genotype.vcf"])
        assert rc_g == 0
        res = {
            "config": f"tools/synth.py {cfg}: {n} alignments x {n_sv} SVs ({mix}, {n_chrom} chromosomes, seed {seed})"
                      + (f"; genotyper on the first {vcf_rows} VCF rows" if vcf_rows else ""),
            "n_aln": n, "vcf_rows": vcf_rows,
            "filter_s": round(t_f, 1), "filter_rc": rc_f, "genotype_s": round(t_g, 1), "genotype_rc": rc_g, "genotype_stdout": so,
            "sha256_json": _sha_file(pre + "_informative_aln.json"), "json_bytes": os.path.getsize(pre + "_informative_aln.json"),
            "sha256_vcf": _sha_file(pre + "_genotype.vcf"),
            "max_rss_mb_children": max(rss_f, rss_g),
            "host": "build container, 1 core of an Intel Xeon @ 2.1 GHz, Python 3.10.12",
            "alignments_per_s": int(n / t_f),
            "how": "python3 /root/reference/filter-alignments.py / predict-genotype.py on the generated files "
                   "(tests/golden/make_golden.py full: the reference is only ever run in the build container)",
        }
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_full(which=("c2", "c3", "c4slice")):
    """synth/c2_full.json, c3_full.json (BASELINE configs[1], configs[2] whole) and c4slice_full.json (the configs[3] graph —
    500 k SVs on 24 chromosomes — with the first million alignments; the genotyper on the first 5 000 VCF rows)."""
    out = f"{HERE}/synth"
    os.makedirs(out, exist_ok=True)
    for w in which:
        if w == "c4slice":
            res = _full_case("c4", n_aln=1_000_000, vcf_rows=5000)
        else:
            res = _full_case(w)
        with open(f"{out}/{w}_full.json", "w") as fh:
            json.dump(res, fh, indent=1)
        print(w, {k: res[k] for k in ("filter_s", "genotype_s", "sha256_json", "sha256_vcf")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["testdir", "quirks", "unicode", "lik", "lik_boundary", "vcf", "synth", "realshape", "utf8order", "nosv", "vcffuzz", "graphfuzz", "dover", "longpath", "contigs", "longtail",
                             "blanks", "fuzz7"]          # (hg002shape: minutes and 7 GB of scratch — by name)
    for w in which:
        if w.startswith("full"):                   # full | full:c2,c3,c4slice
            make_full(tuple(w.split(":")[1].split(",")) if ":" in w else ("c2", "c3", "c4slice"))
        else:
            globals()["make_" + w]()
