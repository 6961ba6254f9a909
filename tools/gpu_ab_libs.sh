#!/bin/bash
# A/B of library builds on the GPU box: kernel time of the 4096-batch bootstrap (even-rotation and mixing instantiation) with
# each library given, plus the output checksums (equal checksums = bit-equal outputs on the seeded inputs).
# usage: bash tools/gpu_ab_libs.sh <tag> <lib.so> ...      (MODES="pbs pbsu" BATCH=4096 REPS=5 by default)
set -o pipefail
TAG=$1; shift
mkdir -p gpurun_out
LOG=gpurun_out/abl_$TAG.log
: > $LOG
for L in "$@"; do
  for M in ${MODES:-pbs pbsu}; do
    echo -n "$L $M " >> $LOG
    SPF_HIP_LIBRARY=$PWD/$L timeout -k 10 120 python3 tools/kernel_bench.py $M ${BATCH:-4096} ${REPS:-5} 2>&1 | tail -1 >> $LOG || { echo "FAILED" >> $LOG; exit 1; }
  done
done
cat $LOG
