#!/bin/bash
# which copy engine do the pool's D2H copies use under different runtime settings?
export TMPDIR=/tmp
run() { # name, env...
  local name=$1; shift
  rm -rf gpurun_out/pe_$name
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pe_$name -- python3 tools/pool_bench.py 1024 > gpurun_out/pe_$name.log 2>&1
  echo "== $name: $(tail -1 gpurun_out/pe_$name.log | grep -o '"circuit_bootstraps_per_s": [0-9.]*')"
  f=$(ls gpurun_out/pe_$name/*/*kernel_stats.csv | head -1)
  grep -i "copyBuffer\|blind_rotate2p2\|ks_gemm" $f | cut -d, -f1-4 | cut -c1-120
}
run base A=1
run blit0 GPU_FORCE_BLIT_COPY_SIZE=0
run sdma1 HSA_ENABLE_SDMA=1
