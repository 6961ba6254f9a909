#!/bin/bash
# circuit-bootstrap tail: parity (tests/test_gpu_cbs_tail.py) and the trace / scheme-switch kernel times of bench.py's
# circuit_bootstrap leg per switch.  usage: bash tools/gpu_trace_ab.sh "<VAR=VAL>" ...
set -o pipefail
for V in "$@"; do
  echo "== $V"
  env $V timeout -k 10 400 python -m pytest tests/test_gpu_cbs_tail.py -x -q -m gpu 2>&1 | tail -2 || exit 1
  env $V timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-counters --no-extras --with-cbs 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)['circuit_bootstrap']
        print('   cbs ms_per_batch', d['ms_per_batch'], 'trace ms', d['trace_roofline']['kernel_ms'], 'frac', d['trace_roofline']['frac'], 'ss ms', d['scheme_switch_roofline']['kernel_ms'])
" || exit 1
done
