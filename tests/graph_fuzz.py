"""Random GRAPHS (not only random lines): hazard-prone chromosome names ("1" inside "11", "chr1" inside "chr11"), insertion nodes
".1" / ".10" / ".11", links with several SVs and both alleles, links present in both reading directions, palindromic links, links
between chromosomes, nodes with more links than a node record holds inline; random walks over them (forwards, backwards, with jumps
the edge table does not know, with revisited nodes, up to 70 nodes), overlap margins on both sides of the 100 bp rule.  TEST
INFRASTRUCTURE: tests/test_gpu_parity.py compares the HIP path with the oracles on these cases."""
import random


def make_case(seed, n_lines=1500):
    """-> (edges dict as in *_svs_edges.json, alt node name -> length, list of GAF lines (str, newline-terminated))"""
    rng = random.Random(seed)
    chroms = rng.sample(["1", "11", "2", "12", "21", "chr1", "chr11", "X", "chrX", "chr1_KI270706v1_random", "MT"], rng.choice((2, 3, 4)))
    ref, length, alt_len = {}, {}, {}
    for c in chroms:
        n = rng.randint(4, 28)
        cuts = sorted(rng.sample(range(2, 60000), n - 1))
        starts = [1] + [x + 1 for x in cuts]
        ends = cuts + [cuts[-1] + rng.randint(30, 3000)]
        ref[c] = [f"{c}:{s}-{e}" for s, e in zip(starts, ends)]
        for nm, s, e in zip(ref[c], starts, ends):
            length[nm] = e - s + 1
    edges = {}
    sv_n = [0]

    def sv(c, kind):
        sv_n[0] += 1
        a, b = rng.randint(1, 90000), rng.randint(1, 90000)
        if kind == "BND":
            c2 = rng.choice(chroms)
            return f"{c}:BND-" + rng.choice((f"{a}[{c2}:{b}[", f"{a}]{c2}:{b}]", f"[{c2}:{b}[{a}", f"]{c2}:{b}]{a}")) + ("" if rng.random() < 0.9 else f"_{sv_n[0]}")
        if kind == "INS":
            return f"{c}:INS-{a}-{sv_n[0] % 12 + 1}"
        return f"{c}:{kind}-{a}-{a + b}"

    def add(l, sl, r, sr, ents):
        edges.setdefault("@".join((l, sl, r, sr)), []).extend(ents)

    all_nodes = []
    for c in chroms:
        nodes = ref[c]
        all_nodes += nodes
        for i in range(len(nodes) - 1):
            ents = [[sv(c, rng.choice(("DEL", "INS", "INV"))), 0] for _ in range(rng.choice((1, 1, 1, 2, 3, 5)))]
            if rng.random() < 0.9:
                add(nodes[i], "+", nodes[i + 1], "+", ents)
            if rng.random() < 0.15:                                   # the same link in the other reading direction, its own SVs
                add(nodes[i + 1], "-", nodes[i], "-", [[sv(c, "INV"), rng.randint(0, 1)]])
            if i + 2 < len(nodes) and rng.random() < 0.35:            # a deletion's alt link
                add(nodes[i], "+", nodes[i + 2], "+", [[sv(c, "DEL"), 1]] + ([[sv(c, "DEL"), 1]] if rng.random() < 0.2 else []))
            if rng.random() < 0.4:                                    # insertion nodes c:pos.cnt (pos = start of the next node), .1 / .10 hazards
                pos = int(nodes[i + 1].split(":")[-1].split("-")[0])
                for cnt in rng.sample((1, 2, 10, 11, 3), rng.choice((1, 1, 2, 3))):
                    an = f"{c}:{pos}.{cnt}"
                    if an in alt_len:
                        continue
                    alt_len[an] = rng.randint(50, 900)
                    length[an] = alt_len[an]
                    s_id = sv(c, "INS")
                    add(nodes[i], "+", an, "+", [[s_id, 1]])
                    add(an, "+", nodes[i + 1], "+", [[s_id, 1]])
                    all_nodes.append(an)
            if rng.random() < 0.25:                                   # inversion links
                j = rng.randrange(len(nodes))
                add(nodes[i], "+", nodes[j], "-", [[sv(c, "INV"), 1]])
                if rng.random() < 0.5:
                    add(nodes[j], "-", nodes[i + 1], "+", [[sv(c, "INV"), 1]])
            if rng.random() < 0.1:                                    # a palindromic link: it is its own reverse
                add(nodes[i], "+", nodes[i], "-", [[sv(c, "INV"), 1]])
        for _ in range(rng.randint(0, 4)):                             # links between chromosomes
            c2 = rng.choice(chroms)
            add(rng.choice(nodes), rng.choice("+-"), rng.choice(ref[c2]), rng.choice("+-"), [[sv(c, "BND"), rng.randint(0, 1)]])
    hub = rng.choice(all_nodes)                                        # a node with more links than its record holds inline
    for _ in range(7):
        add(hub, rng.choice("+-"), rng.choice(all_nodes), rng.choice("+-"), [[sv(hub.split(":")[0], "BND"), rng.randint(0, 1)]])
    keys = list(edges)
    succ = {}
    for k in keys:
        l, sl, r, sr = k.split("@")
        succ.setdefault((l, sl), []).append((r, sr))
        succ.setdefault((r, "-" if sr == "+" else "+"), []).append((l, "-" if sl == "+" else "+"))   # walked the other way
    lines = []
    for i in range(n_lines):
        k = rng.choice((1, 2, 2, 3, 3, 4, 5, 6, 8, 12, 20, 40, 63, 64, 65, 70)) if rng.random() < 0.3 else rng.randint(1, 7)
        cur = (rng.choice(all_nodes), rng.choice("+-"))
        walk = [cur]
        while len(walk) < k:
            r = rng.random()
            if r < 0.85 and cur in succ:
                cur = rng.choice(succ[cur])
            elif r < 0.93:
                cur = (rng.choice(all_nodes), rng.choice("+-"))       # a jump the edge table does not know
            elif r < 0.97 and len(walk) >= 2:
                cur = rng.choice(walk)                                # back to a node of the path (maybe the other way round)
                if rng.random() < 0.5:
                    cur = (cur[0], "-" if cur[1] == "+" else "+")
            else:
                c0 = rng.choice(chroms)
                cur = (f"{c0}:{rng.randint(1, 99999)}-{rng.randint(100000, 199999)}", "+")   # a reference-form name the graph does not have
            walk.append(cur)
        tot = sum(length.get(n, int(n.split("-")[-1]) - int(n.split(":")[-1].split("-")[0]) + 1 if "-" in n.split(":")[-1] else 0) for n, _ in walk)
        path = "".join((">" if s == "+" else "<") + n for n, s in walk)
        ts = rng.choice((0, 0, 50, 99, 100, 101, 400, rng.randint(0, 1200)))
        back = rng.choice((0, 0, 50, 99, 100, 101, 400, rng.randint(0, 1200)))
        tlen = max(tot + rng.choice((0, 0, 0, -7, 13)), 1)
        te = max(tlen - back, ts + 1, 1)
        tags = rng.choice(("tp:A:P\tcm:i:5\ts1:i:50\ts2:i:0\tdv:f:0.01", "tp:A:P", "NM:i:3\tid:f:0.97\ttp:A:P", "tp:A:P\tcg:Z:50M2D40M"))
        lines.append(f"r{seed}_{i}\t{tot + 40}\t3\t{tot + 20}\t{rng.choice('+-')}\t{path}\t{tlen}\t{ts}\t{te}\t{max(te - ts - 3, 1)}\t{max(te - ts, 1)}\t{rng.randint(0, 60)}\t{tags}\n")
    return edges, alt_len, lines
