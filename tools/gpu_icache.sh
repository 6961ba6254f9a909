#!/bin/bash
# instruction-cache counters of one kernel_bench run: bash tools/gpu_icache.sh <tag> <what> <B> [lib.so]
set -o pipefail
TAG=$1; WHAT=$2; B=$3; LIB=${4:-}
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$PWD/gpurun_out/ic_$TAG
mkdir -p $OUT
[ -n "$LIB" ] && export SPF_HIP_LIBRARY=$PWD/$LIB
[ -f gpurun_out/avail_counters.txt ] || rocprofv3 --list-avail > gpurun_out/avail_counters.txt 2>&1 || true
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH --output-format csv -d $OUT/pmc1 -- python3 tools/kernel_bench.py $WHAT $B 3 > $OUT/pmc1.log 2>&1 || { tail -5 $OUT/pmc1.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_EXP --output-format csv -d $OUT/pmc2 -- python3 tools/kernel_bench.py $WHAT $B 3 > $OUT/pmc2.log 2>&1 || { tail -5 $OUT/pmc2.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc1", "pmc2"):
    for f in glob.glob(f"{out}/{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if max(len(v) for v in c.values()) < 2: continue
    print(k, {n: f"{sum(v)/len(v):.4g}" for n, v in c.items()})
PY
