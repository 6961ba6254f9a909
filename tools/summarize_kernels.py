#!/usr/bin/env python3
"""Condense a tools/profile_kernels.sh output tree into profiles/<tag>_kernels_summary.md: one row per (kernel, grid size)
of every case, with the roofline figure that bounds it computed from the algorithmic work stated here.

usage: tools/summarize_kernels.py gpurun_out/kprof_<tag> <tag>
Counters are per dispatch, summed over the chip.  FETCH_SIZE (KB) is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B
requests at 64 B); WRITE_SIZE (KB) is exact.  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles."""
import collections
import csv
import glob
import os
import sys

CLOCK_HZ = 2.39e9          # only where a case has no GRBM_GUI_ACTIVE pass (the cycles of the profiled dispatch = that counter / 8 XCDs)
N_SIMD, N_CU = 1024, 256
FP64_PEAK, HBM_PEAK, I8_PEAK = 78.6e12, 8.0e12, 5.0e15

# algorithmic work per unit: what `achieved` is computed from
GLWE, GGSW = 4096 * 8, 16384 * 16
BSK, AK, SSK = 83492864, 2162688, 491520
# algorithmic HBM bytes of the fp64-bound kernels per launch of B units: key read once + every unit's input and output
ALG_BYTES = {
    "blind_rotate": lambda B: BSK + B * (638 * 8 + GLWE),
    "cbs_trace_kernel": lambda B: AK + B * (GLWE + 4 * GLWE),
    "scheme_switch_kernel": lambda B: SSK + B * (4 * GLWE + GGSW),
}
WORK = {
    # kernel prefix: (bound, unit, work per unit, how many units a workgroup holds)
    "cmux_kernel": ("hbm", "CMUX", GGSW + 3 * GLWE, None),
    "cmux4_kernel": ("hbm", "CMUX", GGSW + 3 * GLWE, 1),
    "ks_gemm_lds_kernel": ("mfma", "keyswitch", 2.0 * 12288 * 638 * 8, None),
    "cbs_trace_kernel": ("fp64", "circuit bootstrap", 4 * 11 * (8 * 51200 + 12 * 8192 + 6 * 6144 + 2 * 8192), None),
    "scheme_switch_kernel": ("fp64", "circuit bootstrap", 4 * (19 * 51200 + 30 * 8192 + 17 * 6144 + 2 * 8192), None),
    "blind_rotate": ("fp64", "PBS", 263.5e6, None),
}


def short(name):
    name = name.split("(")[0]
    for pre in ("void spf::", "spf::", "void "):
        if name.startswith(pre):
            name = name[len(pre):]
    return name.strip()


def main():
    root, tag = sys.argv[1], sys.argv[2]
    lines = [f"# rocprofv3, fixed-batch cases ({tag})", "",
             "Each case is `python3 tools/kernel_bench.py <entry point> <B> 4` (five launches at one batch size, device-resident);",
             "`tools/profile_kernels.sh` runs it once per pass: `--kernel-trace --stats`, then `--pmc` passes, each its own process.",
             "Rows are grouped by (kernel, grid): every dispatch behind a row has the same shape.", ""]
    for case in sorted(glob.glob(os.path.join(root, "*_*"))):
        if not os.path.isdir(case):
            continue
        what, B = os.path.basename(case).rsplit("_", 1)
        B = int(B)
        tr = glob.glob(os.path.join(case, "trace/*/*_kernel_trace.csv"))
        if not tr:
            continue
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            k = short(r["Kernel_Name"])
            if k.startswith(("__amd", "at::", "void at::")) or "elementwise" in k or "distribution" in k or "Cijk" in k:
                continue
            grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
            dur[(k, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        cnt = collections.defaultdict(lambda: collections.defaultdict(list))
        for d in sorted(glob.glob(os.path.join(case, "pmc*"))):
            fs = glob.glob(os.path.join(d, "*/*_counter_collection.csv"))
            if not fs:
                continue
            for r in csv.DictReader(open(fs[0])):
                k = short(r["Kernel_Name"])
                grid = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) if "Grid_Size" in r else 0
                cnt[(k, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        lines += [f"## {what} at B = {B}", "",
                  "| kernel | workgroups | dispatches | avg ms | bound | achieved | frac of peak | HBM bytes (FETCHx2+WRITE) vs algorithmic | VALU busy | LDS busy | MFMA busy | waiting | issue-stalled |",
                  "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
        for (k, grid), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            ms = sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0]   # first dispatch = warm-up
            c = {n: sum(x) / len(x) for n, x in cnt.get((k, grid), {}).items()}
            if not c:   # counter CSVs without a grid column: match by name only
                for (k2, _), cc in cnt.items():
                    if k2 == k:
                        c = {n: sum(x) / len(x) for n, x in cc.items()}
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0 if c.get("GRBM_GUI_ACTIVE") else ms * 1e-3 * CLOCK_HZ
            f = lambda x: "—" if x is None else f"{x:.3f}"
            valu = c["SQ_ACTIVE_INST_VALU"] * 4 / N_SIMD / cyc if "SQ_ACTIVE_INST_VALU" in c else None
            lds = c["SQ_LDS_IDX_ACTIVE"] / N_CU / cyc if "SQ_LDS_IDX_ACTIVE" in c else None
            mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / cyc if c.get("SQ_VALU_MFMA_BUSY_CYCLES") else None
            wait = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
            stall = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
            hbm = c["FETCH_SIZE"] * 1024 * 2 + c.get("WRITE_SIZE", 0) * 1024 if "FETCH_SIZE" in c else None
            bound = ach = frac = alg = "—"
            for pre, (bnd, unit, work, _) in WORK.items():
                if k.startswith(pre):
                    units = B
                    if pre == "cmux4_kernel":
                        units = grid
                    rate = work * units / (ms * 1e-3)
                    peak = {"hbm": HBM_PEAK, "fp64": FP64_PEAK, "mfma": I8_PEAK}[bnd]
                    bound = bnd
                    ach = {"hbm": f"{rate / 1e9:.0f} GB/s", "fp64": f"{rate / 1e12:.2f} TFLOP/s", "mfma": f"{rate / 1e12:.0f} TOP/s"}[bnd]
                    frac = f"{rate / peak:.3f}"
                    if bnd == "hbm":
                        alg = f"{work * units / 1e6:.1f} MB"
                    elif pre in ALG_BYTES:
                        alg = f"{ALG_BYTES[pre](units) / 1e6:.1f} MB"
            hb = "—" if hbm is None else (f"{hbm / 1e6:.1f} MB (fetch x2 {c['FETCH_SIZE'] * 2048 / 1e6:.0f} + write {c.get('WRITE_SIZE', 0) * 1024 / 1e6:.0f})"
                                          + ("" if alg == "—" else f" vs {alg}"))
            lines.append(f"| `{k[:56]}` | {grid} | {len(v)} | {ms:.4f} | {bound} | {ach} | {frac} | {hb} | {f(valu)} | {f(lds)} | {f(mfma)} | {f(wait)} | {f(stall)} |")
        lines.append("")
    os.makedirs("profiles", exist_ok=True)
    open(os.path.join("profiles", f"{tag}_kernels_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
