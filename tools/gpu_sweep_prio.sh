#!/bin/bash
# run every library of a tools/sweep_prio.py sweep: one line per variant (name, schedule, ms per 4096) into gpurun_out/sweep_<which>.log
# usage: bash tools/gpu_sweep_prio.sh EVEN|MIX <list file written by sweep_prio.py, copied into the repo> [reps]
set -o pipefail
WHICH=$1; LIST=$2; REPS=${3:-7}
MODE=pbs; [ "$WHICH" = MIX ] && MODE=pbsu
mkdir -p gpurun_out
LOG=gpurun_out/sweep_$WHICH.log
: > $LOG
for BASE in cur cur; do
  echo -n "base $BASE " >> $LOG
  SPF_HIP_LIBRARY=$PWD/tools/bin/libspf_$BASE.so timeout -k 10 120 python3 tools/kernel_bench.py $MODE 4096 $REPS 2>&1 | tail -1 | sed 's/.*"ms_per_call": \([0-9.]*\).*/\1/' >> $LOG
done
while read NAME SCHED; do
  echo -n "$NAME $SCHED " >> $LOG
  SPF_HIP_LIBRARY=$PWD/tools/bin/libspf_$NAME.so timeout -k 10 120 python3 tools/kernel_bench.py $MODE 4096 $REPS 2>&1 | tail -1 | sed 's/.*"ms_per_call": \([0-9.]*\).*/\1/' >> $LOG || { echo FAILED >> $LOG; exit 1; }
done < $LIST
sort -k3 -n $LOG | head -12
