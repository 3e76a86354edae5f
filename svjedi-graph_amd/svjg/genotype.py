"""Host side of the genotyper: VCF rows -> count slots, svjg_genotype (HIP) -> VCF text.

Mirrors predict-genotype.py's decision_vcf (:89-279) for everything that is file format; the likelihood
(:281-338) and the presence gate (:216) run on the GPU.
"""
import numpy as np

NONE = 0xFFFFFFFF
TYPE_CODE = {"DEL": 0, "INS": 1, "INV": 2, "BND": 3}
GT_TEXT = ("0/0", "0/1", "1/1", "./.")

FORMAT_HEADER = (
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
    '##FORMAT=<ID=DP,Number=1,Type=Float,Description="Total number of informative read alignments across all alleles (after normalization for unbalanced SVs)">\n'
    '##FORMAT=<ID=AD,Number=2,Type=Float,Description="Number of informative read alignments supporting each allele (after normalization by breakpoint number for unbalanced SVs)">\n'
    '##FORMAT=<ID=PL,Number=3,Type=Integer,Description="Phred-scaled likelihood for each genotype">\n'
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"
)


def _info(info, label):
    """predict-genotype.py:77-87 (IndexError when the label is absent, like the reference)."""
    fields = info.split(";")
    if fields[0].startswith(label + "="):
        return info.split(label + "=")[1].split(";")[0]
    if fields[-1].startswith(label + "="):
        return info.split(";" + label + "=")[1]
    return info.split(";" + label + "=")[1].split(";")[0]


def row_key(chrom, pos, alt, info, ins_seen):
    """(svtype, sv_id, length) of one VCF row — predict-genotype.py:118-211."""
    svtype = ""
    if "SVTYPE" in info:
        svtype = info.split("SVTYPE=")[1]
        if not info.split(";")[-1].startswith("SVTYPE="):
            svtype = svtype.split(";")[0]
    end = _info(info, "END") if svtype not in ("BND", "INS") else None
    if svtype in ("DEL", "INV"):
        return svtype, "%s:%s-%s-%s" % (chrom, svtype, pos, end), int(end) - int(pos)
    if svtype == "INS":
        n = ins_seen[pos] = ins_seen.get(pos, 0) + 1        # keyed by POS only, shared by all chromosomes
        return svtype, "%s:INS-%s-%d" % (chrom, pos, n), len(alt)
    if svtype == "BND":
        for br in ("[", "]"):
            if br in alt:
                parts = [x for x in alt.split(br) if x]
                if ":" in parts[1]:
                    return svtype, "%s:BND-%s%s%s%s" % (chrom, pos, br, parts[1], br), 50
                return svtype, "%s:BND-%s%s%s%s" % (chrom, br, parts[0], br, pos), 50
        return svtype, "wrong_format", 50
    return svtype, "unsupported_type", 0


class VcfRows:
    """Input VCF split into header output lines and data rows with their count slot."""

    def __init__(self, vcf_path, slot_of, slot_is_presence=False):
        self.items = []          # ("h", text) or ("r", row index)
        self.prefix = []         # text before "\tGT:DP:AD:PL" for each data row
        types, slots, oks = [], [], []
        ins_seen = {}
        with open(vcf_path) as fh:
            for line in fh:
                if line.startswith("##FORMAT"):
                    continue
                if line.startswith("##"):
                    self.items.append(("h", line))
                    continue
                if line.startswith("#C"):
                    self.items.append(("h", FORMAT_HEADER))
                    continue
                text = line.rstrip("\n")
                cols = text.split("\t")
                chrom, pos, _id, _ref, alt, _q, _f, info, *_rest = cols
                svtype, key, length = row_key(chrom, pos, alt, info, ins_seen)
                code = TYPE_CODE.get(svtype)
                types.append(code if code is not None else 0)
                oks.append((3 if slot_is_presence else 1) if (code is not None and abs(length) >= 50) else 0)
                slots.append(slot_of.get(key, NONE))
                # the first eight columns as they stand (columns 1..8 cannot hold the terminator, so the split of the
                # stripped line has the same first eight as the reference's split of the raw one)
                self.prefix.append(text if len(cols) <= 8 else "\t".join(cols[:8]))
                self.items.append(("r", len(types) - 1))
        self.sv_type = np.array(types, dtype=np.uint8)
        self.slot = np.array(slots, dtype=np.uint32)
        self.ok = np.array(oks, dtype=np.uint8)


def _fmt_counts(svtype_code, ref, alt):
    """AD and DP text after the in-place normalisation of predict-genotype.py:327-338 (Python formatting)."""
    c = [ref, alt]
    if svtype_code == 0 and ref > 0:
        c[0] = round(ref / 2, 1)
    elif svtype_code == 1 and alt > 0:
        c[1] = round(alt / 2, 1)
    return str(round(sum(c), 3)), "%s,%s" % (c[0], c[1])


def exact_pl(svtype_code, ref, alt, err):
    """The three PL integers of one row with the reference's own arithmetic (predict-genotype.py:284-323): double products, Decimal
    sums at precision 28, log10 of the exact binomial coefficient, truncation.  For the rows the kernel flags as lying within 1e-6 of
    an integer boundary (svjg_genotype_boundary): there libm's log10 of a big integer — not necessarily the correctly rounded
    value the kernel uses — could decide the integer."""
    import math
    from decimal import Decimal
    c1, c2 = ref, alt
    if svtype_code == 0 and ref > 0:
        c1 = round(ref / 2, 1)
    elif svtype_code == 1 and alt > 0:
        c2 = round(alt / 2, 1)
    rc1, rc2 = int(round(c1, 0)), int(round(c2, 0))
    l_ok, l_err, l_half = math.log10(1 - err), math.log10(err), math.log10(1 / 2)
    comb = Decimal(math.log10(math.comb(rc1 + rc2, rc1)))
    liks = (Decimal(c1 * l_ok) + Decimal(c2 * l_err), Decimal((c1 + c2) * l_half), Decimal(c2 * l_ok) + Decimal(c1 * l_err))
    return [int(-10 * (x + comb)) for x in liks]


def apply_boundary_guard(ctx, rows, pl, raw, done, err):
    """-> pl with the flagged rows recomputed by exact_pl (a copy only if a row is flagged), number of flagged rows"""
    flags = ctx.boundary_flags(len(rows.sv_type))
    idx = np.flatnonzero(flags & (np.asarray(done) != 0))
    if len(idx) == 0:
        return pl, 0
    pl = np.array(pl, dtype=np.int64)
    for r in idx:
        pl[r] = exact_pl(int(rows.sv_type[r]), int(raw[r, 0]), int(raw[r, 1]), err)
    return pl, len(idx)


def write_vcf(out_path, rows, gt, pl, raw, done):
    n_done = 0
    with open(out_path, "w") as out:
        for kind, v in rows.items:
            if kind == "h":
                out.write(v)
                continue
            if done[v]:
                n_done += 1
                dp, ad = _fmt_counts(int(rows.sv_type[v]), int(raw[v, 0]), int(raw[v, 1]))
                tail = "%s:%s:%s:%d,%d,%d" % (GT_TEXT[gt[v]], dp, ad, pl[v, 0], pl[v, 1], pl[v, 2])
            else:
                tail = "./.:0:0,0:.,.,."
            out.write(rows.prefix[v] + "\tGT:DP:AD:PL\t" + tail + "\n")
    return n_done


def open_rows(vcf_path, slot_of, slot_is_presence=False):
    """The rows of the VCF with their count slots: the native reader (libsvjg_host, svjg_vcf_load) for ordinary files, the
    Python rows above (the semantics, and what raises the reference's exceptions) for everything it declines.
    slot_of: dict key -> slot, or a list of keys (slot = index; a repeated key: the last one wins, like json.load).
    SVJG_PY_VCF=1 forces the Python rows."""
    import os
    from . import capi
    if not os.environ.get("SVJG_PY_VCF"):
        if isinstance(slot_of, dict):
            rows = capi.vcf_load_native(vcf_path, slot_of.keys(), np.fromiter(slot_of.values(), dtype=np.uint32, count=len(slot_of)), slot_is_presence)
        else:
            rows = capi.vcf_load_native(vcf_path, slot_of, None, slot_is_presence)
        if rows is not None:
            return rows
    if not isinstance(slot_of, dict):
        slot_of = {k: i for i, k in enumerate(slot_of)}
    return VcfRows(vcf_path, slot_of, slot_is_presence)


def genotype_with_counts(ctx, vcf_path, slot_of, out_path, min_support=3, err=0.00005, slot_is_presence=False):
    """Counts already live in the context (fused path, or set_counts): parse, run the kernel, write."""
    rows = open_rows(vcf_path, slot_of, slot_is_presence)
    min_support = max(0, int(min_support))                       # (a negative threshold: `sum(nbAln) >= minNbAln` always holds, predict-genotype.py:310)
    bad_err = not (0.0 < float(err) < 1.0)                       # math.log10(e) / math.log10(1 - e) raise in likelihood(), i.e. only once a row gets there
    gt, pl, raw, done = ctx.genotype(rows.sv_type, rows.slot, rows.ok, min_support, 0.5 if bad_err else err, reuse_outputs=True)   # views: written out right away
    if bad_err and np.asarray(done).any():
        raise ValueError("math domain error")                   # (predict-genotype.py:295-297: the reference dies at the first genotyped row)
    pl, _ = apply_boundary_guard(ctx, rows, pl, raw, done, err)
    if isinstance(rows, VcfRows):
        return write_vcf(out_path, rows, gt, pl, raw, done)
    try:
        return rows.write(out_path, gt, pl, raw, done)
    finally:
        rows.close()


def run(json_path, vcf_path, out_path, min_support=3, err=0.00005, device=0):
    """predict-genotype.py main(): counts come from the informative-alignment JSON."""
    from . import capi, filter as flt
    got = flt.read_handoff(json_path)                            # left by our filter-alignments.py for exactly this file, else None
    if got is None:
        got = capi.count_informative_json(json_path)             # len() of the two lists of every key (:219-226)
    keys, counts = got                                           # (a repeated key: the last one wins, like json.load — open_rows)
    ctx = capi.Context(device)
    try:
        ctx.alloc_counts(len(keys))
        ctx.set_counts(counts)
        n = genotype_with_counts(ctx, vcf_path, list(keys), out_path, min_support, err, slot_is_presence=True)
    finally:
        ctx.close()
    print("Genotyped svs: " + str(n))
    return n
