#!/bin/bash
# SQ counters of blind-rotation variants for the experiments log: kernel trace + two PMC passes per variant
# (each pass its own process; --pmc never beside a trace domain).  usage: [KFILTER=cbs_trace BENCH_EXTRA=--with-cbs] bash tools/gpu_counters_ab.sh <tag> "<VAR=VAL>" ...
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
CMD="python3 bench.py --steps 2 --warmup 1 --batch 4096 --no-cpu-baseline --no-extras --no-live-counters ${BENCH_EXTRA:-}"
export KFILTER=${KFILTER:-blind_rotate}
mkdir -p gpurun_out
LOG=gpurun_out/cnt_$TAG.log
: > $LOG
for V in "$@"; do
  N=$(echo "$V" | tr -c 'A-Za-z0-9\n' '_')
  OUT=$PWD/gpurun_out/cnt_${TAG}_$N
  mkdir -p $OUT
  # a variant may set several variables ("A=1 B=2"): all of them are exported for its passes and ALL are unset afterwards
  # (rocprofv3 stays the launched program and python3 the program behind `--`: no env / shell hop under the profiler)
  for KV in $V; do export "$KV"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { echo "$V: rocprofv3 pass failed" >> $LOG; exit 1; }
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1 || { echo "$V: rocprofv3 pass failed" >> $LOG; exit 1; }
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1 || { echo "$V: rocprofv3 pass failed" >> $LOG; exit 1; }
  for KV in $V; do unset "${KV%%=*}"; done
  python3 - "$OUT" "$V" >> $LOG <<'PY'
import csv, glob, os, sys, collections
out, v = sys.argv[1], sys.argv[2]
ms = None
for r in csv.DictReader(open(glob.glob(out + "/trace/*/*_kernel_stats.csv")[0])):
    if os.environ["KFILTER"] in r["Name"]:
        ms, name = float(r["AverageNs"]) / 1e6, r["Name"].split("(")[0].replace("void spf::", "")
acc = collections.defaultdict(list)
for d in ("pmc1", "pmc2"):
    for r in csv.DictReader(open(glob.glob(f"{out}/{d}/*/*_counter_collection.csv")[0])):
        if os.environ["KFILTER"] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(x) / len(x) for k, x in acc.items()}
cyc = ms * 1e-3 * 2.39e9
print(f"{v} | {name} | {ms:.3f} ms | INSTS_VALU {c['SQ_INSTS_VALU']:.4g} | INSTS_LDS {c['SQ_INSTS_LDS']:.4g} | ACTIVE_INST_VALU {c['SQ_ACTIVE_INST_VALU']:.4g} "
      f"(busy {c['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc:.3f}) | LDS_IDX_ACTIVE {c['SQ_LDS_IDX_ACTIVE']:.4g} (busy {c['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:.3f}) | "
      f"WAIT_INST_LDS {c['SQ_WAIT_INST_LDS']:.4g} | WAIT_ANY/WAVE_CYCLES {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f} | WAIT_INST_ANY/WAVE_CYCLES {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f} | "
      f"BANK_CONFLICT {c['SQ_LDS_BANK_CONFLICT']:.3g}")
PY
  [ $? -eq 0 ] || { echo "summary of $V failed" >> $LOG; exit 1; }
done
cat $LOG
