#!/bin/bash
# parity of blind_rotate3_kernel variants against the oracle, then A/B timing.  usage: bash tools/gpu_br3.sh <tag> "<VAR=VAL ...>" ...
set -o pipefail
TAG=$1; shift
mkdir -p gpurun_out
for V in "$@"; do
  echo "== parity $V"
  env $V timeout -k 10 400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "config2 or every_workgroup_shape or every_bootstrap_shape or golden" 2>&1 | tail -3 || exit 1
done
bash tools/gpu_ab.sh $TAG "$@"
