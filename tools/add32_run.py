#!/usr/bin/env python3
"""K 32-bit encrypted additions (BASELINE config 3's circuit, mux_circuits ripple_carry_adder) as ONE gate graph, synthetic
ciphertexts and keys, run a few times — the workload for `rocprofv3 --kernel-trace -- python3 tools/add32_run.py [K]`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device memory for the synthetic keys only)

import spf_amd  # noqa: E402
from spf_amd.gate_pool import circuit_jobs_as_one_graph  # noqa: E402
from spf_amd.mux_circuits import ripple_carry_adder  # noqa: E402
from spf_amd.sharding import key_blob_tensors, replicate_keys  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    P = spf_amd.DEFAULT_128
    dev = torch.device("cuda", 0)
    eng = spf_amd.Engine(P, device=0)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(1)
    blobs = key_blob_tensors(eng, dev)
    for which, t in enumerate(blobs):
        if which == 1:
            t.copy_(torch.randint(-(2 ** 63), 2 ** 63 - 1, (t.numel() // 8,), generator=g0, device=dev, dtype=torch.int64).view(torch.uint8))
        else:
            t.copy_((torch.randn(t.numel() // 8, generator=g0, device=dev, dtype=torch.float64) * 2.0 ** 67).view(torch.uint8))
    replicate_keys(eng, blobs, None, src=0)
    adder = ripple_carry_adder(32, 32, False)
    cts = np.random.default_rng(3).integers(0, 1 << 64, size=(K, 64, P.glwe_words), dtype=np.uint64)
    g, _ = circuit_jobs_as_one_graph(eng, adder, cts)
    g.run()
    for _ in range(3):
        t0 = time.perf_counter()
        g.run()
        print(f"run: {(time.perf_counter() - t0) * 1e3:.3f} ms", g.stats())
    g.close()


if __name__ == "__main__":
    main()
