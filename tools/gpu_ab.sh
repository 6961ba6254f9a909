#!/bin/bash
# A/B timing of blind-rotation variants on the GPU box: each variant is its own process (the library
# reads its SPF_* switches once).  usage: bash tools/gpu_ab.sh <tag> "<VAR=VAL ...>" ["<VAR=VAL ...>" ...]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/ab_$TAG.log
: > $OUT
for V in "$@"; do
  echo "== $V" >> $OUT
  env $V timeout -k 10 240 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-counters 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('   PBS/s', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'])
    elif l: print('   ', l[:300])
" >> $OUT || { echo "   FAILED rc=$?" >> $OUT; exit 1; }
done
cat $OUT
