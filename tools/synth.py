"""Synthetic inputs for the hot path: VCF + variation graph (GFA, _svs_edges.json)
+ GAF.  Input manufacturing only: neither product path nor oracle.

The graph is laid out the way the reference's construct-graph.py lays it out (node names, link keys,
allele-0 entries per breakpoint incl. the BND "mate chromosome" prefix, INS multiplicity counter keyed
by POS only) but in O(M log M); tests/test_synth_vs_reference.py checks it byte for byte against
construct-graph.py itself at small M (build container only).  The reference script is quadratic and
cannot produce the 100 k / 500 k-SV configurations.

Alignments are random walks written by tools/svjg_synth.c (counter-based PRNG per line).
"""
import ctypes
import json
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CONFIGS = {
    # name: (n_aln, n_sv, n_chrom, mix, seed)        BASELINE.json configs[1..3]
    "c2": (1_000_000, 10_000, 1, "del", 20260515 + 1),
    "c3": (10_000_000, 100_000, 4, "mixed", 20260515 + 2),
    "c4": (100_000_000, 500_000, 24, "mixed", 20260515 + 3),
}


def build_lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(HERE, "_build", "libsvjg_synth.so")
    src = os.path.join(HERE, "svjg_synth.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    for fn in (lib.svjg_synth_gaf, lib.svjg_synth_gaf_long, lib.svjg_synth_gaf_reads):
        fn.restype = ctypes.c_long
        fn.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    lib.svjg_synth_gaf_reads.argtypes = lib.svjg_synth_gaf_reads.argtypes + [ctypes.c_void_p]
    _LIB = lib
    return lib


def _sm64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _draws(seed, n, stream):
    with np.errstate(over="ignore"):
        base = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(seed) * np.uint64(1000003) + np.uint64(stream)
        return _sm64(_sm64(base))


_INS_UNIT = "ACGTTGCAAGCT"


_UCSC = ["chr1", "chr2", "chrX", "chr1_KI270706v1_random", "chr14_GL000009v2_random", "chrUn_KI270442v1", "chr17_JH159146v1_alt", "chrUn_GL000195v1",
         "chr3", "chr4", "chr22_KI270879v1_alt", "chr5", "chrY", "chr6", "chr19_KI270938v1_alt", "chr7"]


def chrom_names(n_chrom, style="plain"):
    """plain: chr1 .. chrN; ucsc: GRCh38-style names, contigs of up to 23 bytes among them (node names of up to ~40 bytes)"""
    if style == "ucsc":
        return [_UCSC[c] if c < len(_UCSC) else "chr%d_alt%d" % (c % 22 + 1, c) for c in range(n_chrom)]
    return ["chr%d" % (c + 1) for c in range(n_chrom)]


def make_svs(n_sv, n_chrom, mix, seed, chrom_style="plain"):
    """-> list of SV dicts in VCF order, chromosome names, chromosome lengths."""
    W = 1000 if mix == "del" else 8000
    u_type = _draws(seed, n_sv, 1) % np.uint64(1000)
    u_a = _draws(seed, n_sv, 2)
    u_b = _draws(seed, n_sv, 3)
    u_c = _draws(seed, n_sv, 4)
    types = []
    window = []
    w = -1
    pair_next = False
    for i in range(n_sv):
        if pair_next:
            types.append("INS"); window.append(w); pair_next = False
            continue
        w += 1
        if mix == "del":
            t = "DEL"
        else:
            v = int(u_type[i])
            t = "DEL" if v < 400 else "INS" if v < 800 else "INV" if v < 900 else "BND"
        types.append(t); window.append(w)
        if t == "INS" and int(u_c[i] % np.uint64(100)) < 5:
            pair_next = True
    n_win = w + 1
    chroms = chrom_names(n_chrom, chrom_style)
    first_win = [(c * n_win) // n_chrom for c in range(n_chrom + 1)]
    win_chrom = np.searchsorted(np.array(first_win[1:]), np.arange(n_win), side="right")
    chrom_len = [(first_win[c + 1] - first_win[c] + 2) * W for c in range(n_chrom)]
    svs = []
    for i in range(n_sv):
        wi = window[i]
        c = int(win_chrom[wi])
        j = wi - first_win[c]
        t = types[i]
        if i > 0 and window[i - 1] == wi:                      # second INS of a same-position pair
            pos = svs[-1]["pos"]
        else:
            pos = (j + 1) * W + (0 if mix == "del" else int(u_a[i] % np.uint64(500)))
        sv = {"chrom": c, "pos": pos, "type": t, "idx": i}
        if t == "DEL":
            sv["end"] = pos + 50 + int(u_b[i] % np.uint64(551))
        elif t == "INS":
            sv["alt_len"] = 50 + int(u_b[i] % np.uint64(1951))
        elif t == "INV":
            sv["end"] = pos + 100 + int(u_b[i] % np.uint64(4901))
        else:
            form = int(u_b[i] % np.uint64(4))
            inter = n_chrom > 1 and int((u_b[i] >> np.uint64(8)) % np.uint64(2)) == 0
            mc = int((c + 1 + (u_c[i] >> np.uint64(8)) % np.uint64(n_chrom - 1)) % n_chrom) if inter else c
            nw = first_win[mc + 1] - first_win[mc]
            mj = int((u_c[i] >> np.uint64(20)) % np.uint64(nw))
            sv["form"] = form
            sv["mate_chrom"] = mc
            sv["mate_pos"] = (mj + 1) * W + 6500 + (i % 997)
        svs.append(sv)
    return svs, chroms, chrom_len


def _ins_seq(n):
    return (_INS_UNIT * (n // len(_INS_UNIT) + 1))[:n]


# ---- BASELINE configs[4]'s SHAPE (HG002 GIAB v0.6 Tier1 + 30x ONT): what can be reproduced without minigraph and without the data ----------
# The 24 contigs of GRCh37 with their real lengths (names as in the GIAB v0.6 VCF: no "chr" prefix); ~12.8 k SVs — 5.5 k DEL, 7.3 k INS,
# 50 bp .. 10 kb, small ones most frequent — placed uniformly outside a few SV deserts (centromeres / heterochromatin with their poorly
# mapped flanks, the acrocentric short arms, the tail of Y: approximate hg19 coordinates; Tier1 has no calls there) with a quarter of them
# in clusters of 2..4 within a few kb (tandem-repeat regions); 2 % of the INS share their position with another INS.  With ~240 kb
# between breakpoints a 20 kb read crosses none nine times out of ten: most GAF lines are single-node paths.  The node Y:25 Mbp..59.37 Mbp
# is longer than 2^25 bp.
GRCH37 = [("1", 249250621), ("2", 243199373), ("3", 198022430), ("4", 191154276), ("5", 180915260), ("6", 171115067), ("7", 159138663),
          ("8", 146364022), ("9", 141213431), ("10", 135534747), ("11", 135006516), ("12", 133851895), ("13", 115169878), ("14", 107349540),
          ("15", 102531392), ("16", 90354753), ("17", 81195210), ("18", 78077248), ("19", 59128983), ("20", 63025520), ("21", 48129895),
          ("22", 51304566), ("X", 155270560), ("Y", 59373566)]
_DESERTS = {"1": [(121_000_000, 145_000_000)], "9": [(38_800_000, 71_000_000)], "13": [(0, 19_000_000)], "14": [(0, 19_000_000)],
            "15": [(0, 20_000_000)], "16": [(35_000_000, 46_500_000)], "21": [(0, 14_300_000)], "22": [(0, 16_000_000)],
            "X": [(58_000_000, 62_000_000)], "Y": [(25_000_000, 59_373_566)]}
_WEIGHT = {"X": 0.5, "Y": 0.15}


def make_svs_hg002(seed, n_sv=12_800):
    """-> (SV dicts in VCF order, contig names, contig lengths) for build_graph() / vcf_text()"""
    chroms = [c for c, _ in GRCH37]
    chrom_len = [l for _, l in GRCH37]
    allowed = []
    for c, L in GRCH37:
        iv, at = [], 10_000
        for a, b in sorted(_DESERTS.get(c, [])):
            if a > at:
                iv.append((at, a))
            at = max(at, b)
        if at < L - 30_000:
            iv.append((at, L - 30_000))
        allowed.append(iv)
    room = np.array([sum(b - a for a, b in iv) * _WEIGHT.get(c, 1.0) for (c, _), iv in zip(GRCH37, allowed)])
    per = np.floor(room / room.sum() * n_sv).astype(int)
    per[0] += n_sv - per.sum()
    svs = []
    for c, (iv, n_c) in enumerate(zip(allowed, per)):
        u_pos = _draws(seed, int(n_c), 100 + c)
        u_kind = _draws(seed, int(n_c), 200 + c)
        u_size = _draws(seed, int(n_c), 300 + c)
        u_near = _draws(seed, int(n_c), 400 + c)
        tot = sum(b - a for a, b in iv)
        raw = np.sort((u_pos % np.uint64(tot)).astype(np.int64))
        pos_prev, end_prev, i = 0, 0, 0
        while i < n_c:
            off = int(raw[i])
            for a, b in iv:                                   # offset within the allowed intervals -> coordinate
                if off < b - a:
                    pos = a + off
                    break
                off -= b - a
            members = 1
            if int(u_near[i] % np.uint64(100)) < 9:           # a cluster: this one and 1..3 followers a few hundred bp .. 3 kb apart
                members = 2 + int((u_near[i] >> np.uint64(8)) % np.uint64(3))
            for m in range(members):
                if i >= n_c:
                    break
                if m:
                    pos = max(end_prev, pos_prev) + 60 + int((u_near[i] >> np.uint64(16)) % np.uint64(2940))
                pos = max(pos, pos_prev + 2)
                kind = int(u_kind[i] % np.uint64(1280))
                size = int(50 * 200.0 ** ((int(u_size[i] % np.uint64(1 << 20)) / float(1 << 20)) ** 2))
                sv = {"chrom": c, "pos": pos, "idx": len(svs)}
                if kind < 550:
                    sv["type"] = "DEL"; sv["end"] = pos + size
                    end_prev = sv["end"]
                else:
                    sv["type"] = "INS"; sv["alt_len"] = size
                    end_prev = pos
                svs.append(sv)
                if sv["type"] == "INS" and kind >= 1255 and i + 1 < n_c:     # 2 % of the INS: a second INS at the same position
                    i += 1
                    size2 = int(50 * 200.0 ** ((int(u_size[i] % np.uint64(1 << 20)) / float(1 << 20)) ** 2))
                    svs.append({"chrom": c, "pos": pos, "idx": len(svs), "type": "INS", "alt_len": size2})
                pos_prev = pos
                i += 1
        assert max(sv.get("end", sv["pos"]) for sv in svs if sv["chrom"] == c) < chrom_len[c] - 2
    return svs, chroms, chrom_len


def vcf_text(svs, chroms, chrom_len):
    out = ["##fileformat=VCFv4.2\n", "##source=svjg_synth\n"]
    out += ["##contig=<ID=%s,length=%d>\n" % (c, l) for c, l in zip(chroms, chrom_len)]
    out.append('##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">\n')
    out.append('##INFO=<ID=END,Number=1,Type=Integer,Description="End position">\n')
    out.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
    for sv in svs:
        ch, pos, t = chroms[sv["chrom"]], sv["pos"], sv["type"]
        if t == "DEL":
            alt, info = "<DEL>", "SVTYPE=DEL;END=%d;SVLEN=%d" % (sv["end"], pos - sv["end"])
        elif t == "INS":
            alt, info = _ins_seq(sv["alt_len"]), "SVTYPE=INS;END=%d;SVLEN=%d" % (pos + 1, sv["alt_len"])
        elif t == "INV":
            alt, info = "<INV>", "SVTYPE=INV;END=%d;SVLEN=0" % sv["end"]
        else:
            p = "%s:%d" % (chroms[sv["mate_chrom"]], sv["mate_pos"])
            alt = ("N[%s[" % p, "N]%s]" % p, "]%s]N" % p, "[%s[N" % p)[sv["form"]]
            info = "SVTYPE=BND;SVLEN=0"
        out.append("%s\t%d\tsv%d\tN\t%s\t.\t.\t%s\n" % (ch, pos, sv["idx"], alt, info))
    return "".join(out)


def build_graph(svs, chroms, chrom_len):
    """construct-graph.py:102-554 semantics.  -> dict with edges (link key -> [[sv, allele]...]), GFA text
    pieces and the walk tables."""
    ins_mult = {}
    bk = [dict() for _ in chroms]                    # chrom -> {bkpt: [sv_id...]}
    per_chrom_svs = [[] for _ in chroms]
    sv_ids = []
    for sv in svs:
        c, pos, t = sv["chrom"], sv["pos"], sv["type"]
        if t == "DEL":
            sid = "DEL-%d-%d" % (pos, sv["end"]); pts = [(c, pos), (c, sv["end"])]
        elif t == "INS":
            k = str(pos)
            ins_mult[k] = ins_mult.get(k, 0) + 1
            sv["ins_count"] = ins_mult[k]
            sid = "INS-%d-%d" % (pos, ins_mult[k]); pts = [(c, pos)]
        elif t == "INV":
            sid = "INV-%d-%d" % (pos, sv["end"]); pts = [(c, pos), (c, sv["end"])]
        else:
            mc, p = sv["mate_chrom"], sv["mate_pos"]
            ps = "%s:%d" % (chroms[mc], p)
            f = sv["form"]
            sid = "BND-" + ("%d[%s[" % (pos, ps), "%d]%s]" % (pos, ps), "]%s]%d" % (ps, pos), "[%s[%d" % (ps, pos))[f]
            # breakpoint corrections, construct-graph.py:232-248
            pts = [[(c, pos), (mc, p - 1)], [(c, pos), (mc, p)], [(mc, p), (c, pos - 1)], [(mc, p - 1), (c, pos - 1)]][f]
        sv["sv_id"] = sid
        sv_ids.append(sid)
        seen = set()
        for (bc, bp) in pts:
            if (bc, bp) in seen:
                continue
            seen.add((bc, bp))
            assert 1 < bp < chrom_len[bc] - 1
            bk[bc].setdefault(bp, []).append(sid)
        per_chrom_svs[c].append(sv)

    edges = {}
    gfa = []
    for c, lst in enumerate(per_chrom_svs):
        if lst:
            gfa.append("#%s\t%s\n" % (chroms[c], ";".join(s["sv_id"] for s in lst)))
    node_names, node_len = [], []
    chrom_first = []
    by_end = [dict() for _ in chroms]
    by_start = [dict() for _ in chroms]
    for c, ch in enumerate(chroms):
        chrom_first.append(len(node_names))
        bps = sorted(bk[c])
        starts = [1] + [b + 1 for b in bps]
        ends = bps + [chrom_len[c]]
        if not bps:
            starts, ends = [1], [chrom_len[c]]
        prev = None
        for i, (s, e) in enumerate(zip(starts, ends)):
            nm = "%s:%d-%d" % (ch, s, e)
            by_end[c][e] = len(node_names); by_start[c][s] = len(node_names)
            node_names.append(nm); node_len.append(e - s + 1)
            gfa.append("S\t%s\t*\n" % nm)
            if prev is not None:
                gfa.append("L\t%s\t+\t%s\t+\t0M\n" % (prev, nm))
                edges["%s@+@%s@+" % (prev, nm)] = [["%s:%s" % (ch, sid), 0] for sid in bk[c][bps[i - 1]]]
            prev = nm
        n0 = chrom_first[c]
        gfa.append("P\t%s\t%s\t%s\n" % (ch, ",".join(n + "+" for n in node_names[n0:]),
                                        ",".join("%dM" % l for l in node_len[n0:])))
    chrom_first.append(len(node_names))
    n_ref = len(node_names)

    def add(l, ls, r, rs, key_sv):
        edges.setdefault("%s@%s@%s@%s" % (node_names[l], ls, node_names[r], rs), []).append([key_sv, 1])

    for c, lst in enumerate(per_chrom_svs):
        ch = chroms[c]
        for sv in lst:
            t, pos = sv["type"], sv["pos"]
            ksv = "%s:%s" % (ch, sv["sv_id"])
            if t == "DEL":
                l, r = by_end[c][pos], by_start[c][sv["end"] + 1]
                gfa.append("L\t%s\t+\t%s\t+\t0M\n" % (node_names[l], node_names[r]))
                add(l, "+", r, "+", ksv)
            elif t == "INS":
                nm = "%s:%d.%d" % (ch, pos + 1, sv["ins_count"])
                ni = len(node_names)
                node_names.append(nm); node_len.append(sv["alt_len"])
                l, r = by_end[c][pos], by_start[c][pos + 1]
                gfa.append("S\t%s\t%s\n" % (nm, _ins_seq(sv["alt_len"])))
                gfa.append("L\t%s\t+\t%s\t+\t0M\n" % (node_names[l], nm))
                gfa.append("L\t%s\t+\t%s\t+\t0M\n" % (nm, node_names[r]))
                add(l, "+", ni, "+", ksv); add(ni, "+", r, "+", ksv)
            elif t == "INV":
                l, r = by_end[c][pos], by_start[c][sv["end"] + 1]
                li, ri = by_start[c][pos + 1], by_end[c][sv["end"]]
                gfa.append("L\t%s\t+\t%s\t-\t0M\n" % (node_names[l], node_names[ri]))
                gfa.append("L\t%s\t-\t%s\t+\t0M\n" % (node_names[li], node_names[r]))
                add(l, "+", ri, "-", ksv); add(li, "-", r, "+", ksv)
            else:
                mc, p, f = sv["mate_chrom"], sv["mate_pos"], sv["form"]
                if f == 0:      # t[p[
                    l, ls, r, rs = by_end[c][pos], "+", by_start[mc][p], "+"
                elif f == 1:    # t]p]
                    l, ls, r, rs = by_end[c][pos], "+", by_end[mc][p], "-"
                elif f == 2:    # ]p]t
                    l, ls, r, rs = by_end[mc][p], "+", by_start[c][pos], "+"
                else:           # [p[t
                    l, ls, r, rs = by_start[mc][p], "-", by_start[c][pos], "+"
                gfa.append("L\t%s\t%s\t%s\t%s\t0M\n" % (node_names[l], ls, node_names[r], rs))
                add(l, ls, r, rs, ksv)
    return {"edges": edges, "gfa": gfa, "node_names": node_names, "node_len": node_len, "n_ref": n_ref,
            "sv_keys": ["%s:%s" % (chroms[s["chrom"]], s["sv_id"]) for s in svs]}


def walk_tables(g, seed):
    names = g["node_names"]
    idx = {n: i for i, n in enumerate(names)}
    sv_index = {k: i for i, k in enumerate(g["sv_keys"])}
    arcs = [[] for _ in range(2 * len(names))]
    for key, ents in g["edges"].items():
        l, ls, r, rs = key.split("@")
        a = idx[l] * 2 + (ls == "-"); b = idx[r] * 2 + (rs == "-")
        alt = [e for e in ents if e[1] == 1]
        sv = sv_index[alt[0][0]] if alt else -1
        arcs[a].append((b, sv))
        arcs[b ^ 1].append((a ^ 1, sv))
    ptr = np.zeros(2 * len(names) + 1, dtype=np.uint32)
    ptr[1:] = np.cumsum([len(a) for a in arcs])
    to = np.array([b for a in arcs for (b, _) in a], dtype=np.uint32)
    sv = np.array([s for a in arcs for (_, s) in a], dtype=np.int32)
    blob = "".join(names).encode()
    off = np.zeros(len(names) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(n) for n in names])
    u = _draws(seed, len(g["sv_keys"]), 9) % np.uint64(4)
    gt = np.where(u == 0, 0, np.where(u == 3, 2, 1)).astype(np.uint8)    # 0/0 .25, 0/1 .5, 1/1 .25
    return {"blob": blob, "off": off, "len": np.array(g["node_len"], dtype=np.uint32), "n_ref": g["n_ref"],
            "ptr": ptr, "to": to, "sv": sv, "gt": gt}


def save_tables(tab, prefix):
    """the walk tables of generate() -> {prefix}_walk.npz (another process can then write lines of the same stream)"""
    np.savez(prefix + "_walk.npz", blob=np.frombuffer(tab["blob"], dtype=np.uint8), n_ref=np.array([tab["n_ref"]]),
             **{k: tab[k] for k in ("off", "len", "ptr", "to", "sv", "gt")})


def load_tables(prefix):
    z = np.load(prefix + "_walk.npz")
    tab = {k: np.ascontiguousarray(z[k]) for k in ("off", "len", "ptr", "to", "sv", "gt")}
    tab["blob"] = z["blob"].tobytes()
    tab["n_ref"] = int(z["n_ref"][0])
    return tab


def gaf_bytes(tab, seed, first, n, threads=8, shape="short"):
    """GAF text for lines [first, first+n) as one numpy uint8 array.  shape "long": long-read shaped lines (svjg_synth_gaf_long: read
    names of sequencers, paths long-tailed to 200 nodes, cg:Z: strings on a third of the lines)."""
    lib = build_lib()
    fn = lib.svjg_synth_gaf_long if shape == "long" else lib.svjg_synth_gaf_reads if shape == "reads" else lib.svjg_synth_gaf
    extra = ()
    if shape == "reads":                                        # reads start at genome positions: running sum of the reference nodes' lengths
        cum = np.zeros(tab["n_ref"] + 1, dtype=np.uint64)
        np.cumsum(tab["len"][: tab["n_ref"]], out=cum[1:])
        extra = (cum.ctypes.data,)
    threads = max(1, min(threads, (n + 9999) // 10000))
    step = (n + threads - 1) // threads
    # measurement only (SVJG_SYNTH_HOT=d): reads start in the first 1/d of the reference nodes, so that the records they touch fit the L2
    n_start = max(2, tab["n_ref"] // max(1, int(os.environ.get("SVJG_SYNTH_HOT", "1"))))

    def work(t):
        a = first + t * step
        cnt = max(0, min(step, first + n - a))
        cap = cnt * (2600 if shape == "long" else 700) + 32768
        buf = np.empty(cap, dtype=np.uint8)
        got = fn(tab["blob"], tab["off"].ctypes.data, tab["len"].ctypes.data, n_start,
                                 tab["ptr"].ctypes.data, tab["to"].ctypes.data, tab["sv"].ctypes.data,
                                 tab["gt"].ctypes.data, seed, a, cnt, buf.ctypes.data, cap, *extra)
        assert got >= 0
        return buf[:got]

    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(work, range(threads)))
    return parts[0] if len(parts) == 1 else np.concatenate(parts)


def generate(prefix, n_aln, n_sv, n_chrom, mix, seed, write_gaf=True, threads=8, return_gaf=False, chrom_style="plain", shape="short"):
    """Writes {prefix}.vcf, {prefix}.gfa, {prefix}_svs_edges.json (and {prefix}.gaf)."""
    svs, chroms, chrom_len = make_svs(n_sv, n_chrom, mix, seed, chrom_style)
    with open(prefix + ".vcf", "w") as fh:
        fh.write(vcf_text(svs, chroms, chrom_len))
    g = build_graph(svs, chroms, chrom_len)
    with open(prefix + ".gfa", "w") as fh:
        fh.write("".join(g["gfa"]))
    with open(prefix + "_svs_edges.json", "w") as fh:
        fh.write(json.dumps(g["edges"], sort_keys=True, indent=4))
    tab = walk_tables(g, seed)
    info = {"n_sv": n_sv, "n_nodes": len(g["node_names"]), "n_edge_keys": len(g["edges"]), "tables": tab,
            "chroms": chroms, "chrom_len": chrom_len, "svs": svs}
    if write_gaf or return_gaf:
        buf = gaf_bytes(tab, seed, 0, n_aln, threads, shape)
        info["gaf_bytes"] = int(buf.size)
        if write_gaf:
            buf.tofile(prefix + ".gaf")
        if return_gaf:
            info["gaf"] = buf
    return info


HG002_SEED = 20260515 + 4
HG002_READS = 4_650_000          # 30x of GRCh37's 3.1 Gbp in reads of ~20 kb


def generate_hg002(prefix, n_reads=HG002_READS, seed=HG002_SEED, write_gaf=True, threads=8, return_gaf=False, first=0):
    """BASELINE configs[4]'s shape: writes {prefix}.vcf, {prefix}.gfa, {prefix}_svs_edges.json (and {prefix}.gaf: reads [first, first + n_reads))."""
    svs, chroms, chrom_len = make_svs_hg002(seed)
    with open(prefix + ".vcf", "w") as fh:
        fh.write(vcf_text(svs, chroms, chrom_len))
    g = build_graph(svs, chroms, chrom_len)
    with open(prefix + ".gfa", "w") as fh:
        fh.write("".join(g["gfa"]))
    with open(prefix + "_svs_edges.json", "w") as fh:
        fh.write(json.dumps(g["edges"], sort_keys=True, indent=4))
    tab = walk_tables(g, seed)
    info = {"n_sv": len(svs), "n_nodes": len(g["node_names"]), "n_edge_keys": len(g["edges"]), "tables": tab,
            "chroms": chroms, "chrom_len": chrom_len, "svs": svs}
    if write_gaf or return_gaf:
        buf = gaf_bytes(tab, seed, first, n_reads, threads, "reads")
        info["gaf_bytes"] = int(buf.size)
        if write_gaf:
            buf.tofile(prefix + ".gaf")
        if return_gaf:
            info["gaf"] = buf
    return info


if __name__ == "__main__":
    import sys
    import time
    name = sys.argv[1]
    n_aln, n_sv, n_chrom, mix, seed = CONFIGS[name]
    if len(sys.argv) > 3:
        n_aln = int(sys.argv[3])
    t = time.time()
    inf = generate(sys.argv[2], n_aln, n_sv, n_chrom, mix, seed)
    print({k: v for k, v in inf.items() if k not in ("tables", "svs", "chroms", "chrom_len")}, "%.1fs" % (time.time() - t))
