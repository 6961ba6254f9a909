#!/bin/bash
# Fixed-batch rocprofv3 profiles of the second-tier kernels (VERDICT r2 task 4): per case a kernel trace and PMC passes,
# each its own process (never --pmc beside a trace domain).  usage: bash tools/profile_kernels.sh <tag>
# Output: gpurun_out/kprof_<tag>/<case>/{trace,pmc1,pmc2,pmc3,pmc4,pmc5}; summarise with tools/summarize_kernels.py.
set -o pipefail
TAG=${1:-r03}
export TMPDIR=/tmp
ROOT=$PWD/gpurun_out/kprof_$TAG
for CASE in ${CASES:-"cmux 4096" "cmux 16384" "cmux 256" "keyswitch 4096" "cbs 4096" "pbsu 4096" "pbs 64" "pbs 256" "pbs 512" "pbsu 512" "pbsu 256"}; do
  set -- $CASE
  OUT=$ROOT/$1_$2
  mkdir -p $OUT
  CMD="python3 tools/kernel_bench.py $1 $2 4"
  TRACE_CMD="python3 tools/kernel_bench.py $1 $2 12"   # more launches for the timing row: the first ones of a process run slower
  echo "trace: $TRACE_CMD" > $OUT/command.txt
  echo "pmc:   $CMD" >> $OUT/command.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE_CMD > $OUT/trace.log 2>&1 || { echo "trace failed: $CASE"; exit 1; }
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1 || exit 1
  if [ "$1" = keyswitch ]; then
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 --output-format csv -d $OUT/pmc5 -- $CMD > $OUT/pmc5.log 2>&1 || exit 1
  fi
  echo "done $CASE: $(tail -1 $OUT/trace.log | cut -c1-120)"
done
