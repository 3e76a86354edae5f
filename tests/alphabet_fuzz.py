"""Mutants of GAF lines over the FULL 7-bit alphabet (0x00..0x7F), for differential tests between restatements that must not share a
reading of Python's int() / float() / str.rstrip() (r06: the C oracle and the product's exact routine both took 0x1C..0x1F for blanks
inside int() / float(), where CPython strips them only in str.rstrip(); the campaigns compared the two with each other).

Used by tests/test_oracle_cross_fuzz.py (C oracle and the host build of the product's exact routine against the Python oracle, which
calls Python's own int() / float()), tests/golden/make_golden.py (`blanks`, `fuzz7`: the same mutants through the reference) and
tests/test_gpu_parity.py.  Test infrastructure only.
"""
import random

INT_COLS = (1, 2, 3, 6, 7, 8, 9, 10, 11)
# every byte int() / float() / rstrip() / split("\t") / the terminator scan could treat specially, and their neighbours
BLANKS = bytes([0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x1C, 0x1D, 0x1E, 0x1F, 0x20, 0x00, 0x7F, 0x08, 0x0E, 0x1B, 0x21])
NUMBERS = (b"0", b"00", b"-0", b"+7", b"-7", b"1_0", b"1__0", b"_1", b"1_", b"", b"+", b"-", b"+-1", b"0x10", b"1e3", b"1.0", b"12",
           b"999999999999999999", b"1000000000000000000", b"9223372036854775807", b"9223372036854775808", b"18446744073709551616",
           b"123456789012345678901234567890", b"-123456789012345678901234567890", b"0" * 30 + b"5", b"1" * 641, b"1" * 4300, b"1" * 4301,
           b"0" * 4301, b"1_" * 4300 + b"1", b" 5", b"5 ", b"\x0c5\x0b", b"\x1c5", b"5\x1f", b"5\x00", b"\x005")
FLOATS = (b"0.9", b".5", b"5.", b".", b"1e5", b"1e", b"e5", b"1e+5", b"1e-5", b"1E5", b"inf", b"-Infinity", b"nan", b"+nan", b"infx", b"1_0.5", b"1._5", b"1_.5",
          b"0.9\x1f", b"\x1c0.9", b" 0.9 ", b"0.9\x0c", b"\x0b0.9", b"0.9\x00", b"0x1p3", b"1" * 5000, b"1" * 400 + b"." + b"5" * 400, b"1e" + b"9" * 30, b"", b"--1", b"1 2")


def _cols(b):
    body = bytes(b).rstrip(b"\r\n")
    return body.split(b"\t"), bytes(b)[len(body):]


def mutate7(line, rng):
    """1-3 edits of one GAF line (bytes with its terminator); every edit may put ANY 7-bit byte anywhere, with a bias towards the ends of
    the nine decimal columns, the id:f: value and the line's end"""
    b = bytearray(line)
    for _ in range(rng.choice((1, 1, 1, 2, 2, 3))):
        op = rng.randrange(12)
        cols, term = _cols(b)
        if op == 0 and len(b) > 1:                                   # any byte over any byte
            b[rng.randrange(len(b))] = rng.randrange(128)
        elif op == 1:                                                # any byte inserted anywhere
            b.insert(rng.randrange(len(b) + 1), rng.randrange(128))
        elif op == 2 and len(b) > 1:                                 # a byte deleted
            del b[rng.randrange(len(b))]
        elif op in (3, 4) and len(cols) >= 12:                       # a blank-like byte in front of / behind / inside a decimal column
            c = rng.choice(INT_COLS)
            ch = bytes([rng.choice(BLANKS)]) if op == 3 else bytes([rng.randrange(128)])
            where = rng.randrange(3)
            v = cols[c]
            cols[c] = ch + v if where == 0 else v + ch if where == 1 else v[:len(v) // 2] + ch + v[len(v) // 2:]
            b = bytearray(b"\t".join(cols) + term)
        elif op == 5 and len(cols) >= 12:                            # a decimal column becomes something int() may or may not take
            cols[rng.choice(INT_COLS)] = rng.choice(NUMBERS)
            b = bytearray(b"\t".join(cols) + term)
        elif op == 6 and len(cols) >= 12:                            # an id:f: tag whose value float() may or may not take
            v = rng.choice(FLOATS)
            if rng.random() < 0.5:
                ch = bytes([rng.choice(BLANKS) if rng.random() < 0.7 else rng.randrange(128)])
                v = ch + v if rng.random() < 0.5 else v + ch
            tag = b"id:f:" + v
            if rng.random() < 0.2 and len(cols) > 12:
                cols.insert(rng.randrange(12, len(cols) + 1), tag)
            else:
                cols.append(tag)
            b = bytearray(b"\t".join(cols) + term)
        elif op == 7:                                                # blank-like bytes at the line's end (str.rstrip's set is the larger one)
            tail = bytes(rng.choice(BLANKS) for _ in range(rng.randrange(1, 4)))
            b = bytearray(b"\t".join(cols) + tail + term)
        elif op == 8 and len(cols) >= 12:                            # alignment coordinates: huge, negative, around the 100 bp rule
            c = rng.choice((6, 7, 8))
            try:
                base = int(cols[c])
            except ValueError:
                base = 0
            cols[c] = str(base + rng.choice((-150, -100, -99, -1, 1, 99, 100, 10 ** 18, -10 ** 18, 10 ** 19, 2 ** 61, 2 ** 63, 2 ** 64, 10 ** 30, -10 ** 30))).encode()
            if rng.random() < 0.5:                                   # ... and another of the three moved by the same amount: the difference stays small
                c2 = rng.choice([x for x in (6, 7, 8) if x != c])
                try:
                    cols[c2] = str(int(cols[c2]) + int(cols[c]) - base).encode()
                except ValueError:
                    pass
            b = bytearray(b"\t".join(cols) + term)
        elif op == 9 and len(cols) >= 12 and cols[5][:1] in (b"<", b">"):   # a path node that is no node of the graph: its length is arithmetic on its name
            marks = [i for i, ch in enumerate(cols[5]) if ch in b"<>"]
            i = rng.choice(marks)
            name = rng.choice((b"1:5-104", b"1:1-" + rng.choice(NUMBERS), b"1:" + rng.choice(NUMBERS) + b"-5", b"1:7", b"1:5-", b"1:-5-9", b"1:5-9-3", b"9:1\x1f-9", b"1:\x1c1-9",
                               b"1: 1-9 ", b"1:1_0-2_0", b"1:1-9999999999999", b"1:1-99999999999999999999", b"1:30001-30500 "))
            cols[5] = cols[5][:i] + rng.choice((b"<", b">")) + name + cols[5][i:]
            b = bytearray(b"\t".join(cols) + term)
        elif op == 10:                                               # another terminator
            b = bytearray(b"\t".join(cols) + rng.choice((b"\n", b"\r\n", b"\r", b"", b"\n\n", b"\x0c\n", b"\x1e\n", b"\x1c\r\n")))
        else:                                                        # a column dropped / doubled
            if len(cols) > 2:
                i = rng.randrange(len(cols))
                if rng.random() < 0.5:
                    del cols[i]
                else:
                    cols.insert(i, cols[i])
                b = bytearray(b"\t".join(cols) + term)
    return bytes(b)


# valid UTF-8 the reference's int() / float() / str.rstrip() treat specially (Unicode decimal digits, blanks) or not at all (letters)
UNI = ["\u0663", "\uff15", "\u0967", "\u00a0", "\u2003", "\u0085", "\u3000", "\u200b", "\u00e9", "\u6f22", "\U0001d7d8", "\u00b2", "\u2460"]


def mutate_utf8(line, rng):
    """one mutation of mutate7's kind, then ONE well-formed non-ASCII character at a place that matters: an end of a decimal column, of an id:f: value,
    of the line, inside the read name, inside a node name"""
    b = mutate7(line, rng) if rng.random() < 0.3 else bytes(line)
    cols, term = _cols(b)
    ch = rng.choice(UNI).encode("utf-8")
    where = rng.randrange(6)
    if where == 0 and len(cols) >= 12:
        c = rng.choice(INT_COLS)
        v = cols[c]
        k = rng.randrange(3)
        cols[c] = ch + v if k == 0 else v + ch if k == 1 else v[:1] + ch + v[1:]
    elif where == 1 and len(cols) >= 12:
        v = rng.choice((b"0.9", b"1", b"1e5", b".5"))
        cols.append(b"id:f:" + (ch + v if rng.random() < 0.5 else v + ch))
    elif where == 2:
        return b"\t".join(cols) + ch + term
    elif where == 3 and cols:
        cols[0] = cols[0][:2] + ch + cols[0][2:]
    elif where == 4 and len(cols) >= 12 and cols[5][:1] in (b"<", b">"):
        p = rng.randrange(1, len(cols[5]) + 1)
        cols[5] = cols[5][:p] + ch + cols[5][p:]
    elif len(cols) >= 12:
        cols[rng.choice(INT_COLS)] = "".join(rng.choice("\u0660\u0661\u0662\u0663\u0669\uff10\uff11\uff19" + "0123456789") for _ in range(rng.randrange(1, 6))).encode("utf-8")
    return b"\t".join(cols) + term


def mutants_utf8(lines, n, seed):
    rng = random.Random(seed)
    seen, out = set(), []
    while len(out) < n:
        m = mutate_utf8(rng.choice(lines), rng)
        if m in seen:
            continue
        seen.add(m)
        out.append(m)
    return out


def mutants(lines, n, seed):
    """n distinct mutants of the given lines"""
    rng = random.Random(seed)
    seen, out = set(), []
    while len(out) < n:
        m = mutate7(rng.choice(lines), rng)
        if m in seen:
            continue
        seen.add(m)
        out.append(m)
    return out


def blank_cases(line):
    """The systematic group: each of 0x09-0x0D, 0x1C-0x1F, 0x20, 0x00, 0x7F in front of and behind each of the nine decimal columns of
    `line` (12+ columns, newline terminated), inside an id:f: value (front, back), and at the line's end -> list of (label, bytes)"""
    cols, term = _cols(line)
    out = []
    for ch in (0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x1C, 0x1D, 0x1E, 0x1F, 0x20, 0x00, 0x7F):
        c1 = bytes([ch])
        for c in INT_COLS:
            for where in ("front", "back"):
                x = list(cols)
                x[c] = c1 + x[c] if where == "front" else x[c] + c1
                out.append((f"col{c}_{where}_{ch:02x}", b"\t".join(x) + term))
        for where in ("front", "back"):
            out.append((f"idf_{where}_{ch:02x}", b"\t".join(cols + [b"id:f:" + (c1 + b"0.9" if where == "front" else b"0.9" + c1)]) + term))
            out.append((f"idf_mid_{where}_{ch:02x}", b"\t".join(cols[:12] + [b"id:f:" + (c1 + b"0.9" if where == "front" else b"0.9" + c1)] + cols[12:]) + term))
        out.append((f"end_{ch:02x}", b"\t".join(cols) + c1 + term))
        out.append((f"end12_{ch:02x}", b"\t".join(cols[:12]) + c1 + term))
    return out


def load_packed(path):
    """golden/blanks/blanks.json, golden/fuzz7/fuzz7.json (tests/golden/make_golden.py: _pack_cases) -> list of (fragment bytes, verdict):
    verdict = ("ok", {sv_id: (n_ref, n_alt)}) or ("died", exception class name), as the REFERENCE decided"""
    import base64
    import hashlib
    import json
    import zlib
    d = json.load(open(path))
    text = zlib.decompress(base64.b64decode(d["text_zlib_b64"]))
    assert hashlib.sha256(text).hexdigest() == d["text_sha256"]
    out, pos = [], 0
    for n, v in zip(d["lengths"], d["verdicts"]):
        out.append((text[pos:pos + n], ("died", v) if isinstance(v, str) else ("ok", {k: tuple(x) for k, x in v.items()})))
        pos += n
    assert pos == len(text)
    return out
