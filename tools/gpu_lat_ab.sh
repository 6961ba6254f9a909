#!/bin/bash
# A/B of library builds on the latency shapes: bootstrap batches of 64 / 256 / 512, CMUX batches of 64 / 256, the 32-bit addition
# graph and one 32 x 32 multiplication graph.  usage: bash tools/gpu_lat_ab.sh <tag> <lib.so> ...
set -o pipefail
TAG=$1; shift
mkdir -p gpurun_out
LOG=gpurun_out/lat_$TAG.log
: > $LOG
for L in "$@"; do
  for C in "pbs 64" "pbs 256" "pbsu 256" "pbs 512" "cmux 64" "cmux 256"; do
    echo -n "$L $C " >> $LOG
    SPF_HIP_LIBRARY=$PWD/$L timeout -k 10 120 python3 tools/kernel_bench.py $C 20 2>&1 | tail -1 | cut -c1-200 >> $LOG || { echo FAILED >> $LOG; exit 1; }
  done
  echo -n "$L add32 " >> $LOG
  SPF_HIP_LIBRARY=$PWD/$L timeout -k 10 200 python3 tools/add32_run.py 1 2>&1 | grep "^run" | tail -1 >> $LOG || exit 1
  echo -n "$L mul32 " >> $LOG
  SPF_HIP_LIBRARY=$PWD/$L timeout -k 10 300 python3 tools/mul32_run.py 1 2>&1 | grep "^run" | tail -1 >> $LOG || exit 1
done
cat $LOG
