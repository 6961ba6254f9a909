#!/bin/bash
# compile the library with --save-temps into /tmp/isa and print register / scratch use of the blind-rotation kernels
set -e
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wall -Wno-unused-function --save-temps $EXTRA -o /tmp/isa/lib.so /root/repo/spf_amd/csrc/spf_hip.hip 2>&1 | grep -v "loop not unrolled\|warning generated" || true
python3 - <<'PY'
import re
s=open('/tmp/isa/spf_hip-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r"\.name:\s+(_ZN3spf\w+)\n((?:.*\n){1,14})", s):
    name, body = m.group(1), m.group(2)
    if "Args" not in name and "kernel" not in name: continue
    g=lambda k: (re.search(k+r":\s+(\d+)", body) or [None,None])[1]
    print(f"{name[:60]:60s} vgpr={g('.vgpr_count')} agpr={g('.agpr_count')} sgpr={g('.sgpr_count')} scratch={g('.private_segment_fixed_size')}")
PY
