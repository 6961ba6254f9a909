#!/bin/bash
# usage: lat_quick.sh tag libs...
set -o pipefail
TAG=$1; shift
LOG=gpurun_out/latq_$TAG.log
: > $LOG
for L in "$@"; do
  for C in "pbs 64" "pbs 256" "pbsu 256"; do
    echo -n "$L $C " >> $LOG
    SPF_HIP_LIBRARY=$PWD/$L timeout -k 10 120 python3 tools/kernel_bench.py $C 20 2>&1 | tail -1 | cut -c1-200 >> $LOG || { echo FAILED >> $LOG; exit 1; }
  done
done
cat $LOG
