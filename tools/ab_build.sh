#!/bin/bash
# Build a variant of the library for an A/B on the GPU box: tools/bin/libspf_<name>.so (git-ignored, travels with gpurun).
# usage: bash tools/ab_build.sh <name> [extra hipcc flags, e.g. -DSPF_AB_X=1]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function "$@" \
  -o tools/bin/libspf_$NAME.so spf_amd/csrc/spf_hip.hip 2>&1 | grep -E "error|warning: .*scratch" || true
ls -la tools/bin/libspf_$NAME.so
