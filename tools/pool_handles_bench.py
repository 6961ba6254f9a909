#!/usr/bin/env python3
"""The by-handle legs of bench.py on their own (random keys straight into the key blobs), for A/B runs:
   SPF_HIP_LIBRARY=... python3 tools/pool_handles_bench.py [pool] [add32] [cbs:T[:wait_us[:seconds]]]     one JSON line per leg
   (cbs:1024:200:2 = only the circuit-bootstrap leg at 1 024 callers, quiet time 200 us, 2 s: what tools/pool_kernel_timeline.py reads)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import spf_amd  # noqa: E402
from tools.add32_by_handles import synthetic_engine  # noqa: E402

P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)
eng = synthetic_engine(P)
legs = sys.argv[1:] or ["pool", "add32"]
if "pool" in legs:
    print(json.dumps(bench._bench_pool_by_handle(eng, P, dev, torch)), flush=True)
for leg in legs:
    if leg.startswith("cbs:"):
        f = leg.split(":")
        print(json.dumps(bench._bench_pool_by_handle(eng, P, dev, torch, thread_counts=(int(f[1]),), cmux_cases=(),
                                                     cbs_wait_us=int(f[2]) if len(f) > 2 else 200,
                                                     seconds=float(f[3]) if len(f) > 3 else 2.0)["circuit_bootstrap"]), flush=True)
if "add32" in legs:
    print(json.dumps(bench._bench_add32_by_handles(eng, P)), flush=True)
