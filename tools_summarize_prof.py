#!/usr/bin/env python3
"""Condense a scripts_profile.sh output directory into profiles/<tag>_summary.md."""
import csv, glob, collections, sys, os
src, tag = sys.argv[1], sys.argv[2]
lines = [f"# rocprofv3 summary {tag}", ""]
ks = glob.glob(os.path.join(src, "trace/*/*_kernel_stats.csv"))
if ks:
    lines += ["## --kernel-trace --stats (python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline)", "",
              "| kernel | calls | avg ms | total ms | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks[0])):
        name = r["Name"]
        if len(name) > 90: name = name[:87] + "..."
        lines.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
    lines.append("")
lines += ["## PMC passes (one rocprofv3 --pmc run per row group; per-dispatch averages, dominant kernel only)", "",
          "| pass | counter | dispatches | avg per dispatch |", "|---|---|---|---|"]
for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
    if not os.path.isdir(d): continue
    fs = glob.glob(os.path.join(d, "*/*_counter_collection.csv"))
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "blind_rotate" in r["Kernel_Name"] or "keyswitch_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        lines.append(f"| {os.path.basename(d)} | {k} {c} | {len(v)} | {sum(v)/len(v):.6g} |")
open(os.path.join("profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
