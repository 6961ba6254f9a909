#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc --save-temps .s file, per basic block.

usage: isa_hist.py file.s kernel_substring [--blocks]
Classes: f64 (v_*_f64), valu32 (other v_*), lds (ds_*), vmem (global_/flat_/buffer_/scratch_), salu (s_*).
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(m):
    if m.startswith("v_"):
        if "_f64" in m and not m.startswith("v_cvt"):
            return "f64"
        if m.startswith("v_cvt") and "f64" in m:
            return "cvt64"
        if m.startswith("v_mfma"):
            return "mfma"
        return "valu32"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if m.startswith("s_waitcnt"):
        return "waitcnt"
    if m.startswith("s_barrier"):
        return "barrier"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and key in l:
            start = i
            break
    assert start is not None, "kernel not found"
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:end]:
        s = l.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        mn = s.split()[0]
        blocks[cur].append((mn, s))
        if mn.startswith(("s_cbranch", "s_branch")) and "--split" in sys.argv:
            sub = blocks.get("_n", 0) if False else None
            cur = cur.split("+")[0] + "+" + str(len(blocks))
            blocks[cur] = []
    total = Counter()
    for b, ins in blocks.items():
        c = Counter(classify(m) for m, _ in ins)
        total.update(c)
        if show_blocks:
            br = [s for m, s in ins if m.startswith(("s_cbranch", "s_branch"))]
            print(f"{b:14s} n={len(ins):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + "  " + "; ".join(x.split(";")[0].strip() for x in br))
    print("TOTAL", dict(total))
    if "--mn" in sys.argv:
        blk = sys.argv[sys.argv.index("--mn") + 1]
        c = Counter(m for m, _ in blocks[blk])
        for k, v in c.most_common():
            print(f"  {k:28s} {v}")


if __name__ == "__main__":
    main()
