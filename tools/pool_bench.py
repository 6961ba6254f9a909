#!/usr/bin/env python3
"""The `evaluation_pool` leg of bench.py on its own (random keys straight into the key blobs), for A/B runs of the pool:
   SPF_HIP_LIBRARY=... python3 tools/pool_bench.py [T ...]      prints one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import spf_amd  # noqa: E402
from spf_amd.sharding import _DevArray  # noqa: E402

P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)
eng = spf_amd.Engine(P, device=0)
g = torch.Generator(device=dev).manual_seed(5)
for which in range(4):
    ptr, nbytes = eng.key_blob(which)
    t = torch.as_tensor(_DevArray(ptr, nbytes), device=dev)
    if which == 1:
        t.copy_(torch.randint(-(2 ** 63), 2 ** 63 - 1, (nbytes // 8,), generator=g, device=dev, dtype=torch.int64).view(torch.uint8))
    else:
        t.copy_((torch.randn(nbytes // 8, generator=g, device=dev, dtype=torch.float64) * 2.0 ** 67).view(torch.uint8))
    torch.cuda.synchronize()  # the blob was filled on torch's stream (spf_key_blob_commit also waits for the device since r05)
    eng.key_blob_commit(which)
threads = tuple(int(x) for x in sys.argv[1:]) or (64, 256, 1024)
print(json.dumps(bench._bench_evaluation_pool(eng, P, dev, torch, thread_counts=threads)))
