#!/bin/bash
# A/B of the gate-graph legs of bench.py (cmux kernel, add32, mul8 / mul32 pools) per switch.  usage: bash tools/gpu_extras_ab.sh "<VAR=VAL ...>" ...
set -o pipefail
for V in "$@"; do
  echo "== $V"
  env $V timeout -k 10 500 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-counters 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('   cmux ms', d['cmux']['kernel_ms'], 'add32 ms', d['add32']['ms_per_graph_run'], 'mul8 pool ms', d['mul8_gate_pool']['ms_per_pool_run'],
              'mul32 pool ms', d['mul32_gate_pool']['ms_per_pool_run'], 'errors', d.get('leg_errors'))
" || exit 1
done
