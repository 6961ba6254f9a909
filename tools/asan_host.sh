#!/bin/bash
# Build and run the host-side sanitizer fuzz (tools/fuzz_host.cpp): the library's translation unit with AddressSanitizer +
# UndefinedBehaviorSanitizer on the HOST code only (GPU ASan / XNACK are not available on this pool), no GPU needed.
# usage: bash tools/asan_host.sh [cases (default 1000000)] [seed]
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
if [ ! -x tools/bin/fuzz_host ] || [ tools/fuzz_host.cpp -nt tools/bin/fuzz_host ] || [ -n "$(find spf_amd/csrc include -newer tools/bin/fuzz_host -type f | head -1)" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-gpu-sanitize \
    -fno-sanitize-recover=undefined -ffp-contract=off -std=c++17 -Wno-unused-function -Wno-unused-result tools/fuzz_host.cpp -o tools/bin/fuzz_host
fi
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 tools/bin/fuzz_host "${1:-1000000}" ${2:-}
