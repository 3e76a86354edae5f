"""Host-side graph tables for the HIP classify kernels.

Turns the reference's two graph inputs —
  * `<prefix>_svs_edges.json`  : link key "L@s@R@s" -> [[sv_id, allele], ...]  (filter-alignments.py:95-98)
  * the alt-node S-lines of the GFA: name -> sequence length                     (filter-alignments.py:103-113)
— into the flat arrays described in include/svjg.h (sorted node table, CSR of directed links with the
forward and reversed dictionary entries pre-merged, chromosome dictionary, count-slot numbering).

No classification happens here; this is table construction only.
"""
import json
import os

import numpy as np

NODE_DT = np.dtype([("key", "<u8"), ("aux", "<u4"), ("row", "<u4")])
EDGE_DT = np.dtype([("right", "<u4"), ("meta", "<u4"), ("h0", "<u4"), ("h1", "<u4")])
LEN_UNKNOWN = 0xFFFFFFFF
NODE_HAZARD = 0x80000000
SENTINEL_KEY = 0xFFFFFFFFFFFFFFFF
GRAPH_ALL_SLOW = 1
GRAPH_DOVER_LIST = 16         # svjg.h: the reference was given -O (d_over is a list there: TypeError at the first candidate link)


class GraphFormatError(ValueError):
    """The edge table / GFA uses node names this implementation cannot represent."""


def load_alt_node_len(gfa_path):
    """alt node name -> len(sequence) for S-lines whose last ':'-field contains '.' (filter-alignments.py:105-113)."""
    out = {}
    with open(gfa_path, "r") as fh:
        for line in fh:
            if not line.startswith("S"):
                continue
            cols = line.split("\t")
            if "." in cols[1].split(":")[-1]:
                out[cols[1]] = len(line.rstrip().split("\t")[2])
    return out


def _canon_uint(s):
    return s.isascii() and s.isdigit() and (s == "0" or s[0] != "0") and len(s) <= 10 and int(s) <= 0xFFFFFFFF


def parse_node_name(name):
    """'chrom:start-end' -> (chrom, start, 0, end) ; 'chrom:pos.cnt' -> (chrom, pos, 1, cnt)."""
    if ":" not in name:
        raise GraphFormatError(f"node name without ':' : {name!r}")
    chrom, coords = name.rsplit(":", 1)
    for sep, kind in (("-", 0), (".", 1)):
        if sep in coords:
            a, _, b = coords.partition(sep)
            if _canon_uint(a) and _canon_uint(b):
                if kind == 1 and int(b) >= 32768:
                    raise GraphFormatError(f"alt node multiplicity too large: {name!r}")
                if kind == 0 and int(b) < int(a):
                    raise GraphFormatError(f"node end < start: {name!r}")
                return chrom, int(a), kind, int(b)
            break
    raise GraphFormatError(f"node name is not 'chrom:start-end' or 'chrom:pos.n': {name!r}")


def fnv1a32(b):
    h = 2166136261
    for c in b:
        h = ((h ^ c) * 16777619) & 0xFFFFFFFF
    return h


def _split_key(key):
    p = key.split("@")
    if len(p) != 4 or p[1] not in "+-" or p[3] not in "+-" or len(p[1]) != 1 or len(p[3]) != 1:
        raise GraphFormatError(f"unsupported link key: {key!r}")
    return p[0], p[1] == "-", p[2], p[3] == "-"


class Graph:
    def __init__(self, edges, alt_len, d_over=100, all_slow=False):
        self.d_over = d_over
        self.flags = GRAPH_ALL_SLOW if all_slow else 0
        # ---- nodes --------------------------------------------------------------------------------
        names = set(alt_len)
        parsed_keys = []
        for key, ents in edges.items():
            if not ents:
                continue                      # a present-but-empty key contributes nothing and triggers no check
            l, sl, r, sr = _split_key(key)
            names.add(l); names.add(r)
            parsed_keys.append((l, sl, r, sr, ents))
        info = {n: parse_node_name(n) for n in names}
        chroms = sorted({v[0] for v in info.values()}, key=lambda s: s.encode())
        if len(chroms) >= 65535:
            raise GraphFormatError("too many chromosomes")
        cidx = {c: i for i, c in enumerate(chroms)}
        keyed = {}
        for n, (c, pos, kind, v2) in info.items():
            k = (cidx[c] << 48) | (pos << 16) | (kind << 15) | (v2 if kind else 0)
            if k in keyed:
                raise GraphFormatError(f"two nodes start at the same coordinate: {keyed[k][0]!r} / {n!r}")
            aux = alt_len.get(n, LEN_UNKNOWN) if kind else v2
            if kind and aux != LEN_UNKNOWN and not (0 <= aux < LEN_UNKNOWN):
                raise GraphFormatError("alt node too long")
            keyed[k] = (n, aux)
        order = sorted(keyed)
        n_nodes = len(order)
        self.node_names = [keyed[k][0] for k in order]
        node_id = {nm: i for i, nm in enumerate(self.node_names)}
        nodes = np.zeros(n_nodes + 1, dtype=NODE_DT)
        nodes["key"][:n_nodes] = np.array(order, dtype=np.uint64)
        nodes["aux"][:n_nodes] = np.array([keyed[k][1] for k in order], dtype=np.uint32)
        nodes["key"][n_nodes] = SENTINEL_KEY
        # ---- chromosomes ----------------------------------------------------------------------------
        cb = [c.encode() for c in chroms]
        self.chroms = chroms
        self.chrom_names = b"".join(cb) + b"\0\0\0\0"
        self.chrom_off = np.zeros(len(cb) + 1, dtype=np.uint32)
        self.chrom_off[1:] = np.cumsum([len(b) for b in cb])
        node_chrom = (nodes["key"][:n_nodes] >> np.uint64(48)).astype(np.int64)
        self.chrom_lo = np.searchsorted(node_chrom, np.arange(len(cb) + 1)).astype(np.uint32)
        self.chrom_hash = self._chrom_hash(cb)
        # ---- directed link table: T'[q] = d[q] ++ d[reverse(q)] ----------------------------------------
        table = {}
        for l, sl, r, sr, ents in parsed_keys:
            a, b = node_id[l], node_id[r]
            q = (a, sl, b, sr)
            rq = (b, not sr, a, not sl)
            table.setdefault(q, [None, None])[0] = ents
            table.setdefault(rq, [None, None])[1] = ents
        qs = sorted(table)
        # ---- count slots: numbered in graph order so that one alignment's SVs sit in neighbouring slots --
        slot_of = {}
        sv_ids = []
        merged = []
        for q in qs:
            f, rv = table[q]
            lst = (f or []) + (rv or [])
            hv = []
            for sv, allele in lst:
                if allele not in (0, 1):
                    raise GraphFormatError(f"allele must be 0 or 1 in the edge table (got {allele!r})")
                s = slot_of.get(sv)
                if s is None:
                    s = slot_of[sv] = len(sv_ids)
                    sv_ids.append(sv)
                hv.append((s << 1) | allele)
            merged.append(hv)
        if len(sv_ids) >= (1 << 30):
            raise GraphFormatError("too many SVs")
        self.sv_ids = sv_ids
        self.slot_of = slot_of
        n_e = len(qs)
        edges_a = np.zeros(max(n_e, 1), dtype=EDGE_DT)
        over = []
        left, right, meta, h0, h1 = [], [], [], [], []      # columns as Python lists, converted once (no per-element numpy stores)
        for q, hv in zip(qs, merged):
            a, sl, b, sr = q
            left.append(a); right.append(b)
            meta.append(int(sl) | (int(sr) << 1) | (len(hv) << 2))
            if len(hv) <= 2:
                h0.append(hv[0]); h1.append(hv[1] if len(hv) > 1 else 0)
            else:
                h0.append(len(over)); h1.append(0)
                over.extend(hv)
        if n_e:
            edges_a["right"][:n_e] = np.array(right, dtype=np.uint32)
            edges_a["meta"][:n_e] = np.array(meta, dtype=np.uint32)
            edges_a["h0"][:n_e] = np.array(h0, dtype=np.uint32)
            edges_a["h1"][:n_e] = np.array(h1, dtype=np.uint32)
        rows = np.zeros(n_nodes + 1, dtype=np.int64)
        if n_e:
            rows[1:] = np.bincount(np.array(left, dtype=np.int64), minlength=n_nodes)
        rows = np.cumsum(rows)
        nodes["row"] = rows.astype(np.uint32)
        self.edges = edges_a
        self.n_edges = n_e
        self.hits = np.array(over if over else [0], dtype=np.uint32)
        self.n_hits = len(over)
        # ---- names that are proper substrings of other names (strand quirk, filter-alignments.py:206) ----
        hazard = self._hazards(info, chroms)
        for nm in hazard:
            nodes["row"][node_id[nm]] |= NODE_HAZARD
        self.n_hazard = len(hazard)
        self.nodes = nodes
        self.n_nodes = n_nodes
        self.n_slots = len(sv_ids)

    @staticmethod
    def _hazards(info, chroms):
        """Node names X for which another node name Y contains X as a proper substring."""
        out = set()
        if any(":" in c for c in chroms):
            return Graph._hazards_general(info, chroms)   # (a ':' inside a contig name, e.g. HLA-DRB1*15:03:01:01: the grouping below does not hold)
        suffix_of = {}                        # chrom -> chroms that end with it (incl. itself)
        for c in chroms:
            suffix_of[c] = [d for d in chroms if d.endswith(c)]
        groups = {}
        for n, (c, pos, kind, v2) in info.items():
            groups.setdefault((pos, kind), []).append((n, c, str(v2)))
        for members in groups.values():
            if len(members) < 2:
                continue
            by_chrom = {}
            for n, c, tail in members:
                by_chrom.setdefault(c, []).append((n, tail))
            for n, c, tail in members:
                for d in suffix_of[c]:
                    for n2, tail2 in by_chrom.get(d, ()):
                        if n2 != n and tail2.startswith(tail):
                            out.add(n)
        return out

    @staticmethod
    def _hazards_general(info, chroms):
        """The same set for any contig names (r05; before, a ':' in a contig name sent every line of the graph to the exact routine).
        A node name X = Cx:Tx holds at least one ':' and its tail Tx none, so wherever X occurs inside a name Y its LAST colon
        meets one of Y's colons: Cx is a suffix of what stands in front of that colon and Tx a prefix of what follows it.  For every
        Y and every colon of Y: the contigs that end there x the prefixes behind it that spell a tail (digits, '-' or '.', digits)."""
        out = set()
        cset = set(chroms)
        clens = sorted({len(cx) for cx in chroms})          # (r06: the contigs that END at a colon are looked up by their distinct lengths —
        for y in info:                                       #  thousands of contigs x every colon of every node name was minutes for an analysis set)
            for c in (i for i, ch in enumerate(y) if ch == ":"):
                head, rest = y[:c], y[c + 1:]
                i = 0
                while i < len(rest) and rest[i].isdigit() and rest[i].isascii():
                    i += 1
                if i == 0 or i >= len(rest) or rest[i] not in "-.":
                    continue
                j = i + 1
                ends = []
                while j < len(rest) and rest[j].isdigit() and rest[j].isascii():
                    j += 1
                    ends.append(j)
                for n in clens:
                    if n > len(head):
                        break
                    cx = head[len(head) - n:]
                    if cx in cset:
                        for e in ends:
                            x = cx + ":" + rest[:e]
                            if x != y and x in info:
                                out.add(x)
        return out

    def __getattr__(self, name):
        if name == "node_names":                # natively loaded graph: spelled out on first use (tests, tools)
            ck = self.nodes["key"][: self.n_nodes]
            out = []
            for k, aux in zip(ck.tolist(), self.nodes["aux"][: self.n_nodes].tolist()):
                c, pos, kind, cnt = k >> 48, (k >> 16) & 0xFFFFFFFF, (k >> 15) & 1, k & 0x7FFF
                out.append(f"{self.chroms[c]}:{pos}.{cnt}" if kind else f"{self.chroms[c]}:{pos}-{aux}")
            self.node_names = out
            return out
        raise AttributeError(name)

    @classmethod
    def from_files(cls, edges_json, gfa, d_over=100, all_slow=False, native=None):
        """native: None = the native loader (libsvjg_host.so: svjg_graph_load) when it recognises the files, else this
        module; False = this module only (it defines the semantics); True = native or GraphFormatError."""
        if native is not False and not os.environ.get("SVJG_PY_GRAPH"):
            from . import capi
            t = capi.graph_load_native(edges_json, gfa)
            if t is not None:
                g = cls.__new__(cls)
                g.d_over = d_over
                g.flags = GRAPH_ALL_SLOW if all_slow else 0
                for k in ("nodes", "n_nodes", "edges", "n_edges", "hits", "n_hits", "chrom_names", "chrom_off", "chrom_lo", "sv_ids", "n_hazard"):
                    setattr(g, k, t[k])
                off = g.chrom_off.tolist()
                cb = [g.chrom_names[off[i]:off[i + 1]] for i in range(len(off) - 1)]
                g.chroms = [b.decode("ascii") for b in cb]
                g.chrom_hash = cls._chrom_hash(cb)
                g.slot_of = {sv: i for i, sv in enumerate(g.sv_ids)}
                g.n_slots = len(g.sv_ids)
                return g
            if native:
                raise GraphFormatError("the native loader does not recognise these files")
        with open(edges_json, "r") as fh:
            edges = json.load(fh)
        return cls(edges, load_alt_node_len(gfa), d_over=d_over, all_slow=all_slow)

    @staticmethod
    def _chrom_hash(cb):
        hsize = 8
        while hsize < 2 * len(cb) + 2:
            hsize *= 2
        tab = np.zeros(hsize, dtype=np.uint32)
        for i, b in enumerate(cb):
            j = fnv1a32(b) & (hsize - 1)
            while tab[j]:
                j = (j + 1) & (hsize - 1)
            tab[j] = i + 1
        return tab
