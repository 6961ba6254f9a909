#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + PMC passes (each its own run, no trace domains beside --pmc).
# usage: bash tools/profile_all.sh <tag> [batch] [extra bench args, e.g. "" to keep the extras legs]
set -o pipefail
TAG=${1:-r02}
BATCH=${2:-4096}
ARGS=${3---no-extras}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the kernel trace times the bench's own default run (20 timed steps behind 5 warm-up steps: the first launches of a process
# run 3-5 % slower); the counter passes, which serialise and slow the kernels anyway, take two steps
TRACE_CMD="python3 bench.py --steps 20 --warmup 5 --batch $BATCH --no-cpu-baseline --no-live-counters $ARGS"
CMD="python3 bench.py --steps 2 --warmup 1 --batch $BATCH --no-cpu-baseline --no-live-counters $ARGS"
echo "trace: $TRACE_CMD" > $OUT/command.txt
echo "pmc:   $CMD" >> $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE_CMD > $OUT/trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
echo "pmc2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1
echo "pmc4 rc=$?"
# matrix-core counters (the int8 keyswitch GEMM is the only MFMA user)
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 --output-format csv -d $OUT/pmc5 -- $CMD > $OUT/pmc5.log 2>&1
echo "pmc5 rc=$?"
find $OUT -name "*.csv" | wc -l
