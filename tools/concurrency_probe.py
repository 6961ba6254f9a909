#!/usr/bin/env python3
"""Do two bootstrap launches on two HIP streams run side by side?  (r05: the pool keeps several batches resident.)
   python3 tools/concurrency_probe.py [B]   prints ms for 1 launch, 2 launches on one stream, 2 / 4 launches on separate streams"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import spf_amd  # noqa: E402
from spf_amd.sharding import _DevArray  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)
eng = spf_amd.Engine(P, device=0)
g = torch.Generator(device=dev).manual_seed(5)
ptr, nbytes = eng.key_blob(0)
t = torch.as_tensor(_DevArray(ptr, nbytes), device=dev)
t.copy_((torch.randn(nbytes // 8, generator=g, device=dev, dtype=torch.float64) * 2.0 ** 67).view(torch.uint8))
torch.cuda.synchronize()
eng.key_blob_commit(0)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
lwe = [torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.lwe0_words), generator=g, device=dev, dtype=torch.int64) for _ in range(4)]
out = [torch.empty((B, P.glwe_words), device=dev, dtype=torch.int64) for _ in range(4)]


def run(assign, reps=5):
    for i, s in assign:
        eng.circuit_bootstrap_pbs_dev(streams[s].cuda_stream, B, lwe[i].data_ptr(), out[i].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i, s in assign:
            eng.circuit_bootstrap_pbs_dev(streams[s].cuda_stream, B, lwe[i].data_ptr(), out[i].data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"B = {B} per launch, kernel {eng.last_blind_rotate_kernel() or '-'}")
print(f"1 launch                      {run([(0, 0)]):8.3f} ms")
print(f"2 launches, one stream        {run([(0, 0), (1, 0)]):8.3f} ms")
print(f"2 launches, two streams       {run([(0, 0), (1, 1)]):8.3f} ms")
print(f"4 launches, four streams      {run([(0, 0), (1, 1), (2, 2), (3, 3)]):8.3f} ms")
print(f"kernel {eng.last_blind_rotate_kernel()}")
