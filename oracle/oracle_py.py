"""CPU oracle (pure Python) for the SVJedi-graph hot path.  TEST INFRASTRUCTURE ONLY.

This is a clean-room restatement, written from the behavioural spec in SURVEY.md
Appendix A/B, of what the reference does in

    filter-alignments.py  (alignment classification, /root/reference/filter-alignments.py:95-175)
    predict-genotype.py   (genotype likelihood + VCF rows, /root/reference/predict-genotype.py:89-346)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, and only as the checker.  The product path (svjedi-graph_amd/) never
imports anything from `oracle/`.

Parity pin: `tests/golden/make_golden.py` runs the *reference scripts themselves*
in the build container and stores their outputs under tests/golden/; the tests in
tests/test_oracle_golden.py check this module (and the C oracle) against those
vectors and against the 40 known-answer rows of the reference's
test-dir/expected_genotype.vcf.

Pure-Python loops: use for small cases only (≈10 k alignments/s).
"""
import json
import math
import re
from decimal import Decimal, getcontext

D_OVER = 100  # filter-alignments.py:56,88 (-O N leaves the list ["N"] there: TypeError at the first comparison)


# ----------------------------------------------------------------------------------------------
# filter-alignments.py
# ----------------------------------------------------------------------------------------------

def load_edges(path):
    """filter-alignments.py:95-98 — link key "L@s@R@s" -> [[sv_id, allele], ...]."""
    with open(path) as fh:
        return json.load(fh)


def load_alt_node_len(gfa_path):
    """filter-alignments.py:103-113 — S-lines whose name's last ':'-field has a '.'."""
    out = {}
    with open(gfa_path) as fh:
        for ln in fh:
            if not ln.startswith("S"):
                continue
            cols = ln.split("\t")
            if "." in cols[1].split(":")[-1]:
                out[cols[1]] = len(ln.rstrip().split("\t")[2])
    return out


def _fields(stripped):
    """filter-alignments.py:184-198 — the int() conversions and the Aid evaluation are kept
    because they decide whether the reference raises."""
    c = stripped.split("\t")
    _qid, qlen, qs, qe = c[:4]
    tid, tlen, ts, te = c[5:9]
    am, alen, aq = c[9:12]
    int(qlen), int(qs), int(qe)
    rec = {"Tid": tid, "Tlen": int(tlen), "Ts": int(ts), "Te": int(te)}
    am, alen, aq = int(am), int(alen), int(aq)
    if "id:f:" in stripped:
        float(stripped.split("id:f:")[-1].split("\t")[0])
    else:
        am / alen
    return rec


def _node_names(path):
    """filter-alignments.py:351-373."""
    if path[0] in "<>":
        return [s for s in re.split(r"[<>]", path) if s]
    return [s[:-1] for s in path.split(",") if s]


def _node_len(name, alt_len):
    """filter-alignments.py:328-349."""
    coords = name.split(":")[-1]
    if "." in coords:
        return alt_len[name]
    parts = coords.split("-")
    return int(parts[1]) - int(parts[0]) + 1


def classify_line(line, edges, alt_len, d_over=D_OVER):
    """One GAF line (with its newline) -> list of (sv_id, allele) in reference append order.
    filter-alignments.py:126-166."""
    rec = _fields(line.rstrip())
    path = rec["Tid"]
    names = _node_names(path)
    if len(names) < 2:
        return []
    strands = []
    for nm in names:  # :203-209 — char before the first *substring* occurrence
        strands.append("+" if path.split(nm)[0][-1] == ">" else "-")
    flip = {"+": "-", "-": "+"}
    hits = []
    for i in range(len(names) - 1):
        ln, ls, rn, rs = names[i], strands[i], names[i + 1], strands[i + 1]
        fwd = "@".join((ln, ls, rn, rs))
        rev = "@".join((rn, flip[rs], ln, flip[ls]))
        for key in (fwd, rev):
            if key not in edges:
                continue
            for sv_id, allele in edges[key]:
                il = names.index(ln)
                ir = names.index(rn)
                # (:269-271 — each side is summed and compared in one expression, left first: under -O, where d_over is the
                #  list argparse made, the left comparison raises TypeError before the right sum is formed)
                left_ok = sum(_node_len(n, alt_len) for n in names[: il + 1]) - rec["Ts"] >= d_over
                right_ok = sum(_node_len(n, alt_len) for n in names[ir:]) - (rec["Tlen"] - rec["Te"] - 1) >= d_over
                if left_ok and right_ok:
                    hits.append((sv_id, allele))
    return hits


def classify(gaf_lines, edges, alt_len, d_over=D_OVER):
    """filter-alignments.py:119-166 — sv_id -> [[ref texts], [alt texts]].  d_over: 100, or what `-O N` leaves there: ["N"]."""
    out = {}
    for line in gaf_lines:
        for sv_id, allele in classify_line(line, edges, alt_len, d_over):
            out.setdefault(sv_id, [[], []])[allele].append(line.split("cg:Z:")[0])
    return out


def dump_informative(d):
    """filter-alignments.py:174-175 — exact file text (no trailing newline)."""
    return json.dumps(d, sort_keys=True, indent=4)


def counts_of(d):
    return {k: (len(v[0]), len(v[1])) for k, v in d.items()}


# ----------------------------------------------------------------------------------------------
# predict-genotype.py
# ----------------------------------------------------------------------------------------------

_GT = ("0/0", "0/1", "1/1")


def likelihood(counts, svtype, min_support, err):
    """predict-genotype.py:281-338.  `counts` is mutated (in-place normalisation) like the
    reference does; returns (GT string, [PL0, PL1, PL2] as strings)."""
    getcontext().prec = 28
    if svtype == "DEL" and counts[0] > 0:
        counts[0] = round(counts[0] / 2, 1)
    elif svtype == "INS" and counts[1] > 0:
        counts[1] = round(counts[1] / 2, 1)
    c1, c2 = counts
    r1, r2 = int(round(c1, 0)), int(round(c2, 0))
    l_ok, l_err, l_half = math.log10(1 - err), math.log10(err), math.log10(1 / 2)
    lik = [
        Decimal(c1 * l_ok) + Decimal(c2 * l_err),
        Decimal((c1 + c2) * l_half),
        Decimal(c2 * l_ok) + Decimal(c1 * l_err),
    ]
    best = max(lik)
    arg = [i for i, v in enumerate(lik) if v == best]
    gt = _GT[arg[0]] if len(arg) == 1 else "./."
    if not sum(counts) >= min_support:
        gt = "./."
    comb = Decimal(math.log10(math.comb(r1 + r2, r1)))
    pl = [str(int(-10 * (v + comb))) for v in lik]
    return gt, pl


def _info_value(info, label):
    """predict-genotype.py:77-87."""
    parts = info.split(";")
    if parts[0].startswith(label + "="):
        return info.split(label + "=")[1].split(";")[0]
    if parts[-1].startswith(label + "="):
        return info.split(";" + label + "=")[1]
    return info.split(";" + label + "=")[1].split(";")[0]


def sv_key_of_row(chrom, pos, alt, info, ins_seen):
    """predict-genotype.py:118-211 -> (svtype, sv_id, length)."""
    if "SVTYPE" in info:
        if info.split(";")[-1].startswith("SVTYPE="):
            svtype = info.split("SVTYPE=")[1]
        else:
            svtype = info.split("SVTYPE=")[1].split(";")[0]
    else:
        svtype = ""
    end = None
    if svtype != "BND" and svtype != "INS":
        end = _info_value(info, "END")
    if svtype == "DEL":
        return svtype, f"{chrom}:DEL-{pos}-{end}", int(end) - int(pos)
    if svtype == "INS":
        ins_seen[pos] = ins_seen.get(pos, 0) + 1
        return svtype, f"{chrom}:INS-{pos}-{ins_seen[pos]}", len(alt)
    if svtype == "INV":
        return svtype, f"{chrom}:INV-{pos}-{end}", int(end) - int(pos)
    if svtype == "BND":
        for br in "[]":
            if br in alt:
                p = [x for x in alt.split(br) if x]
                if ":" in p[1]:
                    return svtype, f"{chrom}:BND-{pos}{br}{p[1]}{br}", 50
                return svtype, f"{chrom}:BND-{br}{p[0]}{br}{pos}", 50
        return svtype, "wrong_format", 50
    return svtype, "unsupported_type", None


def genotype_vcf(vcf_lines, informative, min_support=3, err=0.00005):
    """predict-genotype.py:89-279 -> (output text, number of genotyped SVs)."""
    out = []
    ins_seen = {}
    n_gt = 0
    for line in vcf_lines:
        if line.startswith("##FORMAT"):
            continue
        if line.startswith("##"):
            out.append(line)
            continue
        if line.startswith("#C"):
            out.append('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
            out.append('##FORMAT=<ID=DP,Number=1,Type=Float,Description="Total number of informative read alignments across all alleles (after normalization for unbalanced SVs)">\n')
            out.append('##FORMAT=<ID=AD,Number=2,Type=Float,Description="Number of informative read alignments supporting each allele (after normalization by breakpoint number for unbalanced SVs)">\n')
            out.append('##FORMAT=<ID=PL,Number=3,Type=Integer,Description="Phred-scaled likelihood for each genotype">\n')
            out.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
            continue
        cols = line.rstrip("\n").split("\t")
        chrom, pos, _i, _r, alt, _q, _f, info = cols[:8]
        svtype, key, length = sv_key_of_row(chrom, pos, alt, info, ins_seen)
        if svtype in ("DEL", "INS", "INV", "BND") and key in informative and abs(length) >= 50:
            cnt = [len(informative[key][0]), len(informative[key][1])]
            gt, pl = likelihood(cnt, svtype, min_support, err)
            n_gt += 1
        else:
            cnt, gt, pl = [0, 0], "./.", [".", ".", "."]
        raw = line.split("\t")
        head = line.rstrip("\n") if len(raw) <= 8 else "\t".join(raw[:8])
        out.append(
            f"{head}\tGT:DP:AD:PL\t{gt}:{round(sum(cnt), 3)}:{cnt[0]},{cnt[1]}:{','.join(pl)}\n"
        )
    return "".join(out), n_gt
