#!/bin/bash
# bootstrap kernel time against batch size (which shape runs and how long): usage bash tools/batch_curve.sh B1 B2 ...
set -o pipefail
for B in "$@"; do
  timeout -k 10 100 python3 bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($B, d['roofline']['kernel'], d['roofline']['kernel_ms'], 'PBS/s', d['value'])" || exit 1
done
