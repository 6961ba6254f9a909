#!/usr/bin/env python3
"""Print a window of a rocprofv3 --kernel-trace CSV as a per-stream timeline: tools/trace_window.py <csv> [start_ms] [len_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_first = int(rows[0]["Start_Timestamp"])
t_last = int(rows[-1]["End_Timestamp"])
start = float(sys.argv[2]) if len(sys.argv) > 2 else (t_last - t_first) / 2e6
length = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
t0 = t_first + int(start * 1e6)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s > t0 + length * 1e6:
        continue
    name = r["Kernel_Name"].replace("void spf::", "")[:34]
    print(f"q{r['Queue_Id']:>3} s{r.get('Stream_Id', ''):>3} {(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f}  ({(e - s) / 1e6:7.3f} ms) grid {r['Grid_Size_X']:>7} {name}")
