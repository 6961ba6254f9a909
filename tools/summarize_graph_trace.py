#!/usr/bin/env python3
"""Per-kernel totals of the LAST graph run in a `rocprofv3 --kernel-trace` of tools/mul32_run.py (markdown on stdout).
usage: summarize_graph_trace.py <kernel_trace.csv> <launches per run> [title]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per_run = int(sys.argv[2])
title = sys.argv[3] if len(sys.argv) > 3 else "graph run"
rows = [r for r in rows if "spf::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run = the launches after the last long gap (host-side wait between runs)
gaps = sorted(((int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]), i) for i in range(len(rows) - 1)), reverse=True)
cut = max(i for _, i in gaps[:8] if len(rows) - i - 1 >= per_run * 0.9 and len(rows) - i - 1 <= per_run * 1.2) if gaps else -1
last = rows[cut + 1:]
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6
by = collections.OrderedDict()
for r in last:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spf::", "")
    d = by.setdefault(k, [0, 0.0, []])
    d[0] += 1
    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    d[2].append(int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
busy = sum(v[1] for v in by.values())
print(f"### {title}: {len(last)} launches, {span:.2f} ms from first start to last end, {busy:.2f} ms inside kernels\n")
print("| kernel | launches | total ms | avg µs | workgroups per launch (min / median / max) |")
print("|---|---|---|---|---|")
for k, (n, t, wg) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    wg.sort()
    print(f"| `{k}` | {n} | {t:.3f} | {t / n * 1e3:.1f} | {wg[0]} / {wg[len(wg) // 2]} / {wg[-1]} |")
