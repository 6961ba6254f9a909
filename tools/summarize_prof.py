#!/usr/bin/env python3
"""Condense a tools/profile_all.sh output directory into profiles/<tag>_summary.md (every kernel of the run)
and, for the dominant kernel, profiles/latest_counters.json (what bench.py attaches to its roofline block,
labelled as a stored profile).

usage: tools/summarize_prof.py <gpurun_out/prof_TAG> <TAG> [batch]
HBM traffic is corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KB) under-counts wide coalesced reads
by exactly 2x on gfx950 (128-B requests tallied at 64 B); WRITE_SIZE (KB) is exact.  SQ_* counters are summed
over the chip; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (x4 = cycles)."""
import collections
import csv
import glob
import json
import os
import sys
import time

CLOCK_HZ = 2.39e9      # in-kernel clock under this kernel (s_memtime / s_memrealtime probe)
N_SIMD = 1024
N_CU = 256


def short(name: str) -> str:
    name = name.split("(")[0]
    for pre in ("void spf::", "spf::", "void "):
        if name.startswith(pre):
            name = name[len(pre):]
    return name.strip()


def main():
    src, tag = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    lines = [f"# rocprofv3 summary {tag}", ""]
    kernel_ms = {}
    ks = glob.glob(os.path.join(src, "trace/*/*_kernel_stats.csv"))
    if ks:
        lines += ["## --kernel-trace --stats", "", "| kernel | calls | avg ms | total ms | % |", "|---|---|---|---|---|"]
        for r in csv.DictReader(open(ks[0])):
            kernel_ms[short(r["Name"])] = float(r["AverageNs"]) / 1e6
            lines.append(f"| `{short(r['Name'])[:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | "
                         f"{float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
        lines.append("")
    # the same trace grouped by (kernel, workgroups per launch): the bench launches several kernels at more than one
    # batch size (host-pointer slices of 1024, gate-graph levels), and an average over unlike launches says nothing
    kt = glob.glob(os.path.join(src, "trace/*/*_kernel_trace.csv"))
    if kt:
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(kt[0])):
            k = short(r["Kernel_Name"])
            if k.startswith(("__amd", "at::", "void at::")) or "elementwise" in k or "distribution" in k or "Cijk" in k:
                continue
            wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
            by[(k, wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        lines += ["## --kernel-trace, by kernel AND grid (workgroups per launch)", "",
                  "| kernel | workgroups | launches | avg ms | min ms | max ms |", "|---|---|---|---|---|---|"]
        names = sorted({k for k, _ in by})
        for k in names:
            groups = sorted(((g, v) for (kk, g), v in by.items() if kk == k), key=lambda gv: -sum(gv[1]))
            for g, v in groups[:6]:   # the six heaviest shapes of a kernel
                lines.append(f"| `{k[:70]}` | {g} | {len(v)} | {sum(v)/len(v):.4f} | {min(v):.4f} | {max(v):.4f} |")
            if len(groups) > 6:
                rest = [x for _, v in groups[6:] for x in v]
                lines.append(f"| `{k[:70]}` | ({len(groups) - 6} more shapes) | {len(rest)} | {sum(rest)/len(rest):.4f} | {min(rest):.4f} | {max(rest):.4f} |")
        lines.append("")
    lines += ["## PMC passes (one rocprofv3 --pmc run per pass; per-dispatch averages, summed over the chip)", "",
              "| pass | kernel | counter | dispatches | avg per dispatch |", "|---|---|---|---|---|"]
    allacc = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
        if not os.path.isdir(d):
            continue
        fs = glob.glob(os.path.join(d, "*/*_counter_collection.csv"))
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            k = short(r["Kernel_Name"])
            if k.startswith(("__amd", "at::", "void at::")) or "elementwise" in k or "distribution" in k:
                continue
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            allacc[k][c] = sum(v) / len(v)
            lines.append(f"| {os.path.basename(d)} | `{k[:60]}` | {c} | {len(v)} | {sum(v)/len(v):.6g} |")
    # derived figures per kernel
    lines += ["", "## Derived (per dispatch)", "",
              "| kernel | ms | VALU busy | MFMA busy | LDS array busy | wave-cycles waiting (waitcnt/barrier) | issue-stalled | HBM bytes (FETCHx2 + WRITE) |",
              "|---|---|---|---|---|---|---|---|"]
    derived = {}
    for k, c in allacc.items():
        ms = kernel_ms.get(k)
        if not ms:
            continue
        cyc = ms * 1e-3 * CLOCK_HZ
        d = {"kernel_ms": ms}
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / N_SIMD / cyc
        if "SQ_LDS_IDX_ACTIVE" in c:
            d["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / N_CU / cyc
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            d["wait_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
            d["issue_stall_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            d["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / cyc   # cycles the matrix pipes are occupied
        if "FETCH_SIZE" in c:
            d["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2 + c.get("WRITE_SIZE", 0) * 1024
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                  "SQ_INSTS_VALU_MFMA_I8", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA"):
            if n in c:
                d[n] = c[n]
        derived[k] = d
        f = lambda x: "—" if x is None else f"{x:.3f}"
        lines.append(f"| `{k[:60]}` | {ms:.4f} | {f(d.get('valu_busy_frac'))} | {f(d.get('mfma_busy_frac'))} | {f(d.get('lds_busy_frac'))} | "
                     f"{f(d.get('wait_frac'))} | {f(d.get('issue_stall_frac'))} | " +
                     (f"{d['hbm_bytes_per_launch']:.4g}" if 'hbm_bytes_per_launch' in d else "—") + " |")
    os.makedirs("profiles", exist_ok=True)
    open(os.path.join("profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    dom = [k for k in derived if k.startswith("blind_rotate")]
    if dom:
        k = max(dom, key=lambda x: derived[x]["kernel_ms"])
        # the name bench.py gets from spf_last_blind_rotate_kernel: template arguments without spaces
        lib_name = k.replace(", ", ",").replace(",14,0>", ",14,even>").replace(",14,1>", ",14>")
        out = dict(derived[k], kernel=lib_name, batch=batch, tag=tag, date=time.strftime("%Y-%m-%d"),
                   source=f"profiles/{tag}_summary.md: rocprofv3 --kernel-trace and --pmc passes of `bench.py --steps 2 "
                          f"--warmup 1 --no-cpu-baseline --no-extras --batch {batch}` (separate runs); FETCH_SIZE x2 "
                          "(gfx950 128-B requests tallied at 64 B)")
        json.dump(out, open(os.path.join("profiles", "latest_counters.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
