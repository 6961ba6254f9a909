#!/bin/bash
# latency-kernel check on the GPU box: per-phase stamps (diagnostic library), parity of the shapes, kernel time at B=96
set -o pipefail
SPF_HIP_LIBRARY=$PWD/spf_amd/lib/libspf_stamps.so timeout -k 10 200 python3 bench.py --batch 96 --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "stamps\]" | tail -13 &&
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shape or full_parameter" 2>&1 | tail -2 &&
timeout -k 10 100 python3 bench.py --batch 96 --steps 3 --warmup 1 --no-cpu-baseline --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel'], d['roofline']['kernel_ms'])"
