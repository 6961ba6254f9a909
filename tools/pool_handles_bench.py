#!/usr/bin/env python3
"""The by-handle legs of bench.py on their own (random keys straight into the key blobs), for A/B runs:
   SPF_HIP_LIBRARY=... python3 tools/pool_handles_bench.py [cmux] [cbs] [add32]      prints one JSON line per leg"""
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
if "add32" in legs:
    print(json.dumps(bench._bench_add32_by_handles(eng, P)), flush=True)
