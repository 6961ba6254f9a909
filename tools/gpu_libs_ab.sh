#!/bin/bash
# A/B of library builds on the GPU box: kernel time of the batch-96 bootstrap with each tools/bin/libspf_*.so given
set -o pipefail
for L in "$@"; do
  echo "== $L"
  SPF_HIP_LIBRARY=$PWD/$L timeout -k 10 100 python3 bench.py --batch ${BATCH:-96} --steps 3 --warmup 1 --no-cpu-baseline --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel'], d['roofline']['kernel_ms'])" || exit 1
done
