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
allacc = {}
for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
    if not os.path.isdir(d): continue
    fs = glob.glob(os.path.join(d, "*/*_counter_collection.csv"))
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "blind_rotate" in r["Kernel_Name"] or "keyswitch_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    allacc.update(acc)
    for (k, c), v in sorted(acc.items()):
        lines.append(f"| {os.path.basename(d)} | {k} {c} | {len(v)} | {sum(v)/len(v):.6g} |")
open(os.path.join("profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
# HBM traffic of the dominant kernel per launch, corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE (KB) under-counts wide coalesced reads by exactly 2x on gfx950; WRITE_SIZE (KB) is exact.
import json
tr = {}
for (k, c), v in sorted(allacc.items()):
    if "blind_rotate" in k and c in ("FETCH_SIZE", "WRITE_SIZE"):
        tr[c] = sum(v) / len(v)
if "FETCH_SIZE" in tr:
    out = {"fetch_bytes": tr["FETCH_SIZE"] * 1024 * 2, "write_bytes": tr.get("WRITE_SIZE", 0) * 1024,
           "unit": "bytes per launch (B=4096)", "source": f"profiles/{tag}_summary.md: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
           "separate passes; FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B)"}
    out["total_bytes"] = out["fetch_bytes"] + out["write_bytes"]
    json.dump(out, open(os.path.join("profiles", "latest_traffic.json"), "w"), indent=1)
print("\n".join(lines))
