#!/usr/bin/env python3
"""Static instruction census of one kernel in hipcc's -S output, split at the `; MARK n` comments that
-DSVJG_MARK makes the phase stamps (tick()) of k_classify_main emit, and at basic-block labels.

    tools/isa/build.sh [extra -D flags]       -> /tmp/isa/main.s
    python3 tools/isa/count.py /tmp/isa/main.s [--blocks]
"""
import re
import sys
from collections import Counter, OrderedDict


def kind(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "xlane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path = sys.argv[1]
    blocks = "--blocks" in sys.argv
    seg = "pre"
    segs = OrderedDict()
    blk = None
    per_blk = OrderedDict()
    for line in open(path):
        s = line.strip()
        m = re.match(r";\s*MARK\s+(\S+)", s)
        if m:
            seg = "after MARK " + m.group(1)
            continue
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            blk = m.group(1)
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        k = kind(op)
        segs.setdefault(seg, Counter())[k] += 1
        per_blk.setdefault((seg, blk), Counter())[k] += 1
    cols = ["valu", "xlane", "salu", "lds", "vmem", "smem", "branch", "wait", "other"]
    print(f"{'segment':28s}" + "".join(f"{c:>8s}" for c in cols))
    tot = Counter()
    for sname, c in segs.items():
        print(f"{sname:28s}" + "".join(f"{c[k]:8d}" for k in cols))
        tot.update(c)
    print(f"{'total':28s}" + "".join(f"{tot[k]:8d}" for k in cols))
    if blocks:
        print()
        for (sname, b), c in per_blk.items():
            print(f"{sname:20s} {str(b):12s}" + "".join(f"{c[k]:7d}" for k in cols))


if __name__ == "__main__":
    main()
