#!/usr/bin/env python3
"""Do quarter batches of the keyswitch -> circuit-bootstrap chain, running side by side on their own streams, add up to the rate of
one full batch?  G contexts (own stream, own intermediates, own key replica), each looping the chain over B / G ciphertexts,
against one context over B.  usage: concurrent_chains_probe.py [B] [G] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import spf_amd  # noqa: E402
from spf_amd.sharding import key_blob_tensors, replicate_keys  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)


def engine(seed):
    eng = spf_amd.Engine(P, device=0)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(seed)
    blobs = key_blob_tensors(eng, dev)
    for which, t in enumerate(blobs):
        if which == 1:
            t.copy_(torch.randint(-(2 ** 63), 2 ** 63 - 1, (t.numel() // 8,), generator=g0, device=dev, dtype=torch.int64).view(torch.uint8))
        else:
            t.copy_((torch.randn(t.numel() // 8, generator=g0, device=dev, dtype=torch.float64) * 2.0 ** 67).view(torch.uint8))
    replicate_keys(eng, blobs, None, src=0)
    return eng


def buffers(n):
    return (torch.randint(-(2 ** 63), 2 ** 63 - 1, (n, P.lwe1_words), device=dev, dtype=torch.int64),
            torch.empty((n, P.lwe0_words), device=dev, dtype=torch.int64),
            torch.empty((n, P.cbs_ggsw_complex * 2), device=dev, dtype=torch.float64))


engs = [engine(1) for _ in range(G)]
streams = [torch.cuda.Stream(device=dev) for _ in range(G)]


def run(parts, n_each):
    bufs = [buffers(n_each) for _ in range(parts)]

    def step():
        for i in range(parts):
            a, m, o = bufs[i]
            engs[i].keyswitch_dev(streams[i].cuda_stream, n_each, a.data_ptr(), m.data_ptr())
            engs[i].circuit_bootstrap_dev(streams[i].cuda_stream, n_each, m.data_ptr(), o.data_ptr())

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return parts * n_each * reps / (time.perf_counter() - t0)


one = run(1, B)
many = run(G, B // G)
print(f"B = {B}: one stream {one:.0f} circuit bootstraps/s; {G} streams x {B // G}: {many:.0f}/s = {many / one:.3f} of it "
      f"(blind rotation of a part: {engs[0].last_blind_rotate_kernel()})")
