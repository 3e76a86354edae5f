"""Structure-aware fuzzer for LONG paths (65..216 nodes): the main kernel walks such a line in sub-passes of 64 nodes and counts in its
first sweep while no name has come twice, so what matters is WHERE in the line something happens.  tests/graph_fuzz.py walks at most 70
nodes over graphs of ~100 nodes (a walk comes back to a node inside its first 64), tests/golden/make_fuzz.py makes long paths by
repeating a path: neither reaches a line whose first sub-passes are clean and whose LATER sub-pass meets the event.  Here:

* graphs of >= 2 000 nodes: chromosomes "1" and "11" with the same breakpoints (every node name of "1" is a substring of one of "11":
  the strand quirk of filter-alignments.py:206 — such names take the exact path), "chr2", a contig with a long UCSC name, a
  chromosome of three 40 Mbp nodes (a node too long for the main kernel's 32-bit sums); insertion nodes, deletion links, links in both
  reading directions;
* walks of 65..216 nodes that do NOT come back to a node in their first 64: straight along a chromosome (through insertion nodes and
  over deletion links), forwards or backwards; then ONE late event at a position >= 64 (often right at a sub-pass boundary or the
  line's end): a reference-form name the graph lacks (length positive / NEGATIVE — the lower bound the first sweep counts against is no
  bound then — / of 5 Gbp), a name of 49..60 bytes, a hazard name, a 40 Mbp node, a name that came before (either orientation), ids that
  turn, a jump to another contig, a stretch walked back — or none;
* margins Ts / Te that leave 0..70 nodes at the far end undecided, at the near end too; cg:Z: tails beyond the 8 KB stage;
* `fatal`: the same with an insertion node the GFA lacks (KeyError in the reference as soon as a link with a candidate SV sums over it,
  filter-alignments.py:343-345): one line per run, the error must win.

TEST INFRASTRUCTURE: tests/test_gpu_parity.py (HIP path against both oracles), tests/test_hostsim_parity.py (the exact routine),
tests/golden/make_golden.py (golden/longpath: 40 lines of this generator through the reference)."""
import random

EVENTS = ("none", "unknown_ref", "unknown_ref_negative", "unknown_ref_huge", "long_name", "hazard", "big_node", "revisit", "revisit_flipped",
          "id_turn", "contig_jump", "fold_back", "unknown_ref_twice")


def make_graph(seed):
    """-> (edges dict as in *_svs_edges.json, alt node name -> length, {chrom: [ref node names in order]}, {name: length})"""
    rng = random.Random(seed * 7919 + 13)
    ref, length, alt_len, edges = {}, {}, {}, {}
    sv_n = [0]

    def sv(c, kind, pos):
        sv_n[0] += 1
        if kind == "INS":
            return f"{c}:INS-{pos}-{sv_n[0] % 9 + 1}"
        return f"{c}:{kind}-{pos}-{pos + rng.randint(50, 5000)}"

    def add(l, sl, r, sr, ents):
        edges.setdefault("@".join((l, sl, r, sr)), []).extend(ents)

    shared = None
    for c, n in (("1", 420), ("11", 420), ("chr2", 900), ("chr1_KI270706v1_random", 330), ("chrBig", 3)):
        if c == "chrBig":
            starts, ends = [1, 40000001, 80000001], [40000000, 80000000, 120000000]
        elif c == "11":
            starts, ends = shared                                    # the breakpoints of "1": "1:s-e" is a substring of "11:s-e"
        else:
            cuts = sorted(rng.sample(range(200, 2_000_000), n - 1))
            starts = [1] + [x + 1 for x in cuts]
            ends = cuts + [cuts[-1] + rng.randint(200, 3000)]
            if c == "1":
                shared = (starts, ends)
        nodes = [f"{c}:{s}-{e}" for s, e in zip(starts, ends)]
        ref[c] = nodes
        for nm, s, e in zip(nodes, starts, ends):
            length[nm] = e - s + 1
        for i in range(len(nodes) - 1):
            pos = starts[i + 1]
            ents = [[sv(c, rng.choice(("DEL", "INV")), pos), 0] for _ in range(rng.choice((1, 1, 1, 2, 4)))]
            add(nodes[i], "+", nodes[i + 1], "+", ents)
            if rng.random() < 0.1:
                add(nodes[i + 1], "-", nodes[i], "-", [[sv(c, "INV", pos), rng.randint(0, 1)]])
            if i + 2 < len(nodes) and rng.random() < 0.3:
                add(nodes[i], "+", nodes[i + 2], "+", [[sv(c, "DEL", pos), 1]])
            if rng.random() < 0.3 and c != "chrBig":
                an = f"{c}:{pos}.{rng.choice((1, 1, 2, 3))}"
                if an not in alt_len:
                    alt_len[an] = length[an] = rng.randint(50, 900)
                    s_id = sv(c, "INS", pos)
                    add(nodes[i], "+", an, "+", [[s_id, 1]])
                    add(an, "+", nodes[i + 1], "+", [[s_id, 1]])
    return edges, alt_len, ref, length


def _name_len(n, length):
    if n in length:
        return length[n]
    c = n.split(":")[-1]
    if "." in c:
        return 0                                                     # (an insertion node the GFA lacks: fatal lines only)
    a, b = c.rsplit("-", 1) if not c.startswith("-") else (c, "0")
    return int(b) - int(a) + 1


def make_case(seed, n_lines=60, n_fatal=6, graph_seed=None):
    """-> (edges, alt, lines, fatal): `lines` the reference classifies, `fatal` lines it dies on (one per run)"""
    edges, alt, ref, length = make_graph(seed if graph_seed is None else graph_seed)
    rng = random.Random(seed)
    nxt, ins_after = {}, {}
    for key in edges:
        l, ls, r, rs = key.split("@")
        if ls == "+" and rs == "+":
            nxt.setdefault(l, []).append(r)
    where = {}
    for c, nodes in ref.items():
        for i, n in enumerate(nodes):
            where[n] = (c, i, 1)
    for an in alt:
        c, rest = an.rsplit(":", 1)
        pos = int(rest.split(".")[0])
        i = next(i for i, n in enumerate(ref[c]) if int(n.rsplit(":", 1)[1].split("-")[0]) == pos)
        where[an] = (c, i, 0)                                       # in front of reference node i

    def walk(c, start, k):
        """k nodes straight on from ref[c][start]: through insertion nodes and over deletion links now and then; no node twice"""
        cur, out = ref[c][start], [ref[c][start]]
        while len(out) < k:
            cand = sorted((x for x in nxt.get(cur, []) if where[x][:2] > where[cur][:2] or (where[x][1] == where[cur][1] and where[x][2] > where[cur][2])),
                          key=lambda x: (where[x][1], where[x][2]))
            if not cand:
                break
            r = rng.random()
            ins = [x for x in cand if where[x][2] == 0]
            cur = ins[0] if ins and r < 0.4 else cand[-1] if r > 0.9 else next((x for x in cand if where[x][2] == 1), cand[0])
            out.append(cur)
        return out

    def one(tag, fatal_line):
        c = rng.choice(("chr2", "chr2", "chr2", "11", "chr1_KI270706v1_random"))
        kmax = 150 if c.startswith("chr1_") else 216              # (names of 36 bytes: the path stays inside a stripe)
        k = rng.choice((65, 66, 100, 127, 128, 129, 130, 160, 191, 192, 193, 200, 215, 216))
        k = min(k, kmax)
        w = walk(c, rng.randrange(0, len(ref[c]) - 240), k)
        at = rng.choice((64, 64, 65, 70, 100, 126, 127, 128, 129, 150, 190, 191, 192, len(w) - 1, len(w) - 1, len(w) - 2))
        at = max(64, min(at, len(w) - 1))
        ev = "missing_alt" if fatal_line else rng.choice(EVENTS)
        ori = [">"] * len(w)
        other = rng.choice([x for x in ("chr2", "11", "chr1_KI270706v1_random") if x != c])
        if ev == "unknown_ref":
            w[at] = f"{c}:{rng.randint(3_000_000, 4_000_000)}-{rng.randint(4_000_001, 4_002_000)}"
        elif ev == "unknown_ref_twice":
            w[at] = f"{c}:3000001-3000900"
            if at + 3 < len(w):
                w[at + 3] = w[at]
        elif ev == "unknown_ref_negative":
            w[at] = f"{c}:{rng.randint(5_000_000, 6_000_000)}-{rng.randint(4_000_000, 4_990_000)}"       # int(end) - int(start) + 1 < 0
        elif ev == "unknown_ref_huge":
            w[at] = f"{c}:1-{rng.randint(4_300_000_000, 5_000_000_000)}"
        elif ev == "long_name":
            w[at] = "chrUn_" + "JTFH0100" + "9" * rng.randint(22, 33) + f"v1_decoy:{rng.randint(1, 900)}-{rng.randint(1000, 1999)}"
        elif ev == "hazard":
            w[at] = rng.choice(ref["1"])
        elif ev == "big_node":
            w[at] = ref["chrBig"][rng.randrange(3)]
        elif ev in ("revisit", "revisit_flipped"):
            w[at] = w[rng.choice((0, 1, 30, 63, max(at - 2, 0), max(at - 64, 0)))]
            if ev == "revisit_flipped":
                ori[at] = "<"
        elif ev == "id_turn":
            back = walk(c, rng.randrange(0, 30), len(w) - at)
            if not set(back) & set(w[:at]):
                w[at:] = back[: len(w) - at]
        elif ev == "contig_jump":
            w[at:] = walk(other, rng.randrange(0, 60), len(w) - at)
        elif ev == "fold_back":
            n_back = min(len(w) - at, at)
            w[at:] = list(reversed(w[:at]))[:n_back] + w[at + n_back:]
            for i in range(at, at + n_back):
                ori[i] = "<"
        elif ev == "missing_alt":
            pos = int(w[at].rsplit(":", 1)[1].split("-")[0].split(".")[0])
            w[at] = f"{c}:{pos}.7"                                   # (no insertion of the graph has count 7)
        lens = [_name_len(n, length) for n in w]
        tl = sum(lens)
        rev = rng.random() < 0.5
        if rev:
            w, ori, lens = list(reversed(w)), [("<" if o == ">" else ">") for o in reversed(ori)], list(reversed(lens))
        # margins: the overlap rule asks for 100 bp on the left of a link's left node and on the right of its right node
        front, tail = [0], [0]
        for x in lens:
            front.append(front[-1] + x)
        for x in reversed(lens):
            tail.append(tail[-1] + x)
        und = rng.choice((0, 0, 1, 2, 5, 20, 62, 63, 64, 65, 70))
        und = min(und, len(w) - 2)
        ts = rng.choice((0, 5, 99, 100, 101, max(front[rng.choice((1, 2, 10))] - 50, 0)))
        te_back = max(tail[und] - 90, 0) if und else rng.choice((0, 7, 99, 100, 101))
        tlen = max(tl + rng.choice((0, 0, 0, -7, 13)), 1) if tl > 0 else 1000
        tlen = min(tlen, 999_999_999)
        ts = min(ts, 99_999_999)
        te = min(max(tlen - te_back, 1), 999_999_999)
        path = "".join(o + n for o, n in zip(ori, w))
        tags = rng.choice(("tp:A:P\tcm:i:5\ts1:i:50\ts2:i:0\tdv:f:0.01", "tp:A:P", "tp:A:P\tcg:Z:" + "50M2D40M" * rng.choice((3, 1200))))
        return f"{tag}_{ev}_k{len(w)}a{at}{'r' if rev else 'f'}\t{min(tl + 40, 999_999_999) if tl > 0 else 50000}\t3\t{min(tl + 20, 999_999_990) if tl > 0 else 40000}\t{rng.choice('+-')}\t{path}\t{tlen}\t{ts}\t{te}\t{max(te - ts - 3, 1)}\t{max(te - ts, 1)}\t{rng.randint(0, 60)}\t{tags}\n"

    lines = [one(f"s{seed}_{i}", False) for i in range(n_lines)]
    fatal = [one(f"s{seed}_x{i}", True) for i in range(n_fatal)]
    # short ordinary lines in between, so that long lines meet ordinary passes in their stripes
    mixed = []
    for i, l in enumerate(lines):
        mixed.append(l)
        for j in range(rng.randint(0, 3)):
            c = rng.choice(("chr2", "11", "chr1_KI270706v1_random", "1"))
            w = walk(c, rng.randrange(0, len(ref[c]) - 20), rng.randint(1, 9))
            tl = sum(length[n] for n in w)
            rv = rng.random() < 0.5
            p = "".join(("<" if rv else ">") + n for n in (reversed(w) if rv else w))
            mixed.append(f"s{seed}_{i}_o{j}\t{tl}\t0\t{tl}\t+\t{p}\t{tl}\t{rng.choice((0, 50, 150))}\t{tl - rng.choice((0, 60, 170))}\t{tl}\t{tl}\t60\ttp:A:P\n")
    return edges, alt, mixed, fatal


def make_tail_case(seed, base_lines, n_mut=400):
    """Lines LONGER than the main kernel's 8 KB stage (r04: such a line stays in the main kernel when its columns and path lie in its first
    8 KB and what runs past the stage is a plain tail; the worker walks the tail 4 KB a step, 16 bytes a lane): `base_lines` (bytes, no
    terminator) get tails of 6..40 KB made of tag text in which ONE thing happens at a position chosen to sit on / next to a 16-byte, 64-byte,
    4 KB or 8 KB boundary of the LINE or of the FILE — nothing; a carriage return (the reference's universal newlines end the line there: what
    follows is a line of its own — kept well-formed here); the byte pair "d:" outside a tag, split across a boundary or not; an id:f: tag
    with a plain / a malformed value (the malformed ones are FATAL: returned apart); bytes >= 0x80 (valid UTF-8); a tab-separated tail of many
    short tags; the terminator itself at a boundary; no terminator at all (last line).  -> (accepted text: bytes, [fatal texts: bytes])"""
    rng = random.Random(seed)
    out, fatal = [], []
    pos = 0
    alphabet = b"ACGTacgtMIDNSHP=X0123456789"

    def filler(n):
        return bytes(rng.choice(alphabet) for _ in range(min(n, 64))) * (n // 64 + 1)

    picks = set(rng.sample(range(len(base_lines)), min(n_mut, len(base_lines))))
    for i, l in enumerate(base_lines):
        if i not in picks:
            out.append(l + b"\n")
            pos += len(l) + 1
            continue
        tail_len = rng.choice((6000, 8100, 8192, 9000, 12288, 16384, 20000, 40000, 53376 - len(l), 70000)) + rng.randint(-40, 40)   # (53 376 B: what the one-wave-per-line kernel stages of a line)
        kind = rng.choice(("plain", "plain", "cr", "dcolon", "dcolon_split", "idf_ok", "idf_bad", "utf8", "many_tags", "end_on_boundary"))
        head = l + b"\tcg:Z:"
        body = bytearray(filler(tail_len)[:tail_len])
        # a position whose offset in the line or in the file sits on / next to a boundary
        b = rng.choice((16, 64, 4096, 8192))
        target = rng.choice((b * rng.randint(1, max(1, (len(head) + tail_len) // b)), len(head) + rng.randint(0, tail_len - 1)))
        if rng.random() < 0.5:
            target = target - ((pos + target) % b) + rng.choice((-1, 0, 0, 1))          # on a boundary of the FILE offset
        if rng.random() < 0.25:
            target = 8192 - (pos % 16) + rng.randint(-24, 8)                            # around the END OF THE STAGE of a stripe that begins with this line (seed 271 of r05's campaign: a tag cut there)
        at = min(max(target - len(head), 8), tail_len - 24)
        extra = b""
        if kind == "cr":
            body[at:at + 1] = b"\r"
            # what follows the CR is a line of its own for the reference: make it one it accepts (a short well-formed line)
            rest = b"r\t9\t0\t9\t+\t>nonode\t9\t0\t9\t9\t9\t60"
            body = body[:at + 1] + rest
        elif kind == "dcolon":
            body[at:at + 4] = b"\txd:"[:4]
        elif kind == "dcolon_split":
            body[at:at + 2] = b"d:"
        elif kind == "idf_ok":
            body[at:at + 18] = b"\tid:f:0.9875\tzz:Z:"
        elif kind == "idf_bad":
            body[at:at + 18] = b"\tid:f:0.9x75\tzz:Z:"
        elif kind == "utf8":
            body[at:at + 4] = "éè".encode()
        elif kind == "many_tags":
            for q in range(8, len(body) - 8, rng.choice((40, 97, 513))):
                body[q:q + 1] = b"\t"
        elif kind == "end_on_boundary":
            want_len = ((pos + len(head) + tail_len) // b) * b + rng.choice((-1, 0, 1, 15, 16)) - pos - len(head)
            body = body[:max(16, min(want_len, len(body)))]
        line = head + bytes(body) + extra
        if kind == "idf_bad":
            fatal.append(b"".join(out[-3:]) + line + b"\n")
            out.append(l + b"\n")
            pos += len(l) + 1
            continue
        out.append(line + b"\n")
        pos += len(line) + 1
    text = b"".join(out)
    if rng.random() < 0.5:
        text = text[:-1]                                             # the file's last line without a terminator
    return text, fatal


def make_soup(seed, n_long=40, graph_seed=None, terminators=False):
    """Everything at once on ONE graph (make_graph), shuffled into one file so that the main kernel's special cases meet at stripe
    boundaries: long paths with one late event (make_case), the same lines with 6..30 KB tails of tag text in front of their terminator
    (a plain cg:Z: string, a "d:" pair, an id:f: tag with a plain value), runs of 30..120 tiny lines of 28..40 bytes (more lines than a
    stripe's list holds: the stripe is cut at a line start), ordinary short walks, and empty lines' worth of nothing in between.
    -> (edges, alt, text: bytes)"""
    edges, alt, lines, _fatal = make_case(seed, n_long, 0, graph_seed)
    rng = random.Random(seed * 31 + 5)
    out = []
    alphabet = "ACGT0123456789MIDS="
    some_node = next(iter(edges)).split("@")[0]
    for l in lines:
        r = rng.random()
        body = l[:-1]
        if r < 0.25:                                                  # a tail beyond the 8 KB stage
            n = rng.choice((6000, 8192, 12000, 30000)) + rng.randint(-30, 30)
            tail = "".join(rng.choice(alphabet) for _ in range(64)) * (n // 64 + 1)
            kind = rng.choice(("plain", "plain", "dcolon", "idf"))
            ins = {"plain": "", "dcolon": "\txd:Z:", "idf": "\tid:f:0.9911\tzz:Z:"}[kind]
            at = rng.randint(8, n - 8)
            body = body + "\tcg:Z:" + tail[:at] + ins + tail[at:n]
        out.append(body + "\n")
        if rng.random() < 0.3:                                        # a run of tiny lines: more line starts than a stripe's list holds
            for i in range(rng.randint(30, 120)):
                out.append(f"t{i}\t9\t0\t9\t+\t>{some_node}\t9\t0\t9\t9\t9\t{i % 61}\n" if rng.random() < 0.8 else f"u{i}\t1\t0\t1\t-\t<{some_node}\t1\t0\t1\t1\t1\t0\n")
    rng.shuffle(out)
    if terminators:                                                   # the reference reads in text mode: "\n", "\r\n" and a lone "\r" all end a line (and reach the JSON as "\n")
        out = [l[:-1] + rng.choice(("\n",) * 8 + ("\r\n", "\r\n", "\r")) for l in out]
        for i in range(len(out) - 1):                                 # (a lone "\r" in front of a line that begins with "\n"-less text only: "\r" + "\n..." would be ONE terminator)
            if out[i].endswith("\r") and not out[i].endswith("\r\n") and out[i + 1].startswith("\n"):
                out[i] = out[i][:-1] + "\n"
    return edges, alt, "".join(out).encode()
