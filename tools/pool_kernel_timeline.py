#!/usr/bin/env python3
"""What the GPU does while the pool runs many resident batches: reads a `rocprofv3 --kernel-trace --output-format csv` trace of
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pool_tl -- python3 tools/pool_handles_bench.py cbs:1024
and prints, for the steady part of the run: per kernel the count, the median / p90 duration and the workgroups; per queue the
chain of one batch (keyswitch .. scheme switch) with the gaps between its kernels; the number of blind rotations resident over
time and the share of the CUs their workgroups hold (a two-per-SIMD blind-rotation workgroup owns its CU).
usage: pool_kernel_timeline.py <dir or csv> [n_cu]"""
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np


def find_csv(path):
    if os.path.isfile(path):
        return path
    hits = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
    if not hits:
        raise SystemExit(f"no *kernel_trace.csv under {path}")
    return hits[-1]


def short(name):
    name = name.split("(")[0]
    for tag in ("blind_rotate2p2", "blind_rotate2p", "blind_rotate8", "cbs_trace", "scheme_switch", "ks_gemm", "keyswitch", "gather_rows",
                "cmux", "spin_kernel", "copy_words"):
        if tag in name:
            return tag
    return name.split("<")[0][-40:]


def main():
    path = find_csv(sys.argv[1])
    n_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            wg = max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
            grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r.get("Queue_Id", 0) or 0),
                         grid // wg if grid else 0))
    rows.sort()
    t_first, t_last = rows[0][0], max(r[1] for r in rows)
    # the steady part: the last 60 % of the blind rotations' time span
    br = [r for r in rows if r[2].startswith("blind_rotate")]
    if not br:
        raise SystemExit("no blind rotation in the trace")
    lo = br[0][0] + (br[-1][1] - br[0][0]) * 4 // 10
    hi = br[-1][0]
    steady = [r for r in rows if lo <= r[0] and r[1] <= hi]
    print(f"{path}: {len(rows)} kernels over {(t_last - t_first) / 1e9:.2f} s; steady window {(hi - lo) / 1e6:.1f} ms, {len(steady)} kernels")
    by = defaultdict(list)
    for s, e, k, q, wgs in steady:
        by[k].append(((e - s) / 1e3, wgs))
    print(f"{'kernel':>18} {'count':>6} {'median us':>10} {'p90 us':>9} {'median wgs':>10} {'busy share':>10}")
    for k, v in sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
        d = np.array([x[0] for x in v])
        print(f"{k:>18} {len(v):6d} {np.median(d):10.1f} {np.percentile(d, 90):9.1f} {int(np.median([x[1] for x in v])):10d} "
              f"{d.sum() * 1e3 / (hi - lo):10.2f}")
    # resident blind rotations and the CUs they hold (per-workgroup CU ownership: 2p / 2p2 one per CU up to n_cu; 8: one per CU)
    ev = []
    for s, e, k, q, wgs in steady:
        if k.startswith("blind_rotate"):
            ev.append((s, +1, wgs))
            ev.append((e, -1, wgs))
    ev.sort()
    t_prev, n_res, cus = lo, 0, 0
    acc_res, acc_cu, hist = 0.0, 0.0, defaultdict(float)
    for t, sgn, wgs in ev:
        dt = t - t_prev
        acc_res += n_res * dt
        acc_cu += min(cus, n_cu) * dt
        hist[n_res] += dt
        n_res += sgn
        cus += sgn * wgs
        t_prev = t
    span = t_prev - lo
    print(f"blind rotations resident: mean {acc_res / span:.2f}; their workgroups hold {acc_cu / span / n_cu:.3f} of {n_cu} CUs on average")
    print("  time share by number resident:", ", ".join(f"{n}: {v / span:.2f}" for n, v in sorted(hist.items())))
    # chains per queue: consecutive kernels of one queue from a keyswitch to the scheme switch
    by_q = defaultdict(list)
    for r in steady:
        by_q[r[3]].append(r)
    chains, gaps = [], defaultdict(list)
    for q, v in by_q.items():
        cur = []
        for r in v:
            if r[2] in ("ks_gemm", "keyswitch") and cur and any(x[2].startswith("blind_rotate") for x in cur):
                chains.append(cur)
                cur = []
            cur.append(r)
        if cur and any(x[2] == "scheme_switch" for x in cur):
            chains.append(cur)
    full = [c for c in chains if any(x[2].startswith("blind_rotate") for x in c) and any(x[2] == "scheme_switch" for x in c)]
    if full:
        life = np.array([(c[-1][1] - c[0][0]) / 1e3 for c in full])
        brd = np.array([next((x[1] - x[0]) / 1e3 for x in c if x[2].startswith("blind_rotate")) for c in full])
        for c in full:
            for a, b in zip(c, c[1:]):
                gaps[f"{a[2]} -> {b[2]}"].append((b[0] - a[1]) / 1e3)
        print(f"{len(full)} whole chains on {len(by_q)} queues: first kernel start to last kernel end median {np.median(life):.0f} us "
              f"(p90 {np.percentile(life, 90):.0f}); the blind rotation in it {np.median(brd):.0f} us")
        for k, v in sorted(gaps.items(), key=lambda kv: -np.median(kv[1])):
            print(f"   gap {k:>34}: median {np.median(v):8.1f} us, p90 {np.percentile(v, 90):8.1f}  ({len(v)})")
        # idle time of a queue between two chains (completion -> callers back -> next batch of that set)
        idle = []
        for q, v in by_q.items():
            mine = sorted((c for c in full if c[0][3] == q), key=lambda c: c[0][0])
            for a, b in zip(mine, mine[1:]):
                idle.append((b[0][0] - a[-1][1]) / 1e3)
        if idle:
            print(f"   a queue between two of its chains: median {np.median(idle):.0f} us idle (p90 {np.percentile(idle, 90):.0f})")


if __name__ == "__main__":
    main()
