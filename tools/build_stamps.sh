#!/bin/bash
# diagnostic library with per-phase s_memtime stamps in blind_rotate2p_kernel: spf_amd/lib/libspf_stamps.so
# use: SPF_HIP_LIBRARY=$PWD/spf_amd/lib/libspf_stamps.so python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function -DSPF_STAMPS $EXTRA \
  -o spf_amd/lib/libspf_stamps.so spf_amd/csrc/spf_hip.hip 2>&1 | grep -E "error|scratch" || true
ls -la spf_amd/lib/libspf_stamps.so
