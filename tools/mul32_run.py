#!/usr/bin/env python3
"""One 32 x 32 encrypted multiplication (BASELINE config 5's circuit) as a gate graph, synthetic ciphertexts and keys,
run a few times — the workload for `rocprofv3 --kernel-trace --stats -- python3 tools/mul32_run.py`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device memory for the synthetic keys only)

import spf_amd  # noqa: E402
from spf_amd import FheCircuit, ValueKind  # noqa: E402
from spf_amd.mux_circuits import GraphBuilder, append_uint_multiply, parse_mux_circuit  # noqa: E402


class _DevArray:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


def main():
    jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    P = spf_amd.DEFAULT_128
    dev = torch.device("cuda", 0)
    eng = spf_amd.Engine(P, device=0)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(1)
    for which in range(4):
        ptr, nbytes = eng.key_blob(which)
        t = torch.as_tensor(_DevArray(ptr, nbytes), device=dev)
        if which == 1:
            t.copy_(torch.randint(-(2 ** 63), 2 ** 63 - 1, (nbytes // 8,), generator=g0, device=dev, dtype=torch.int64).view(torch.uint8))
        else:
            t.copy_((torch.randn(nbytes // 8, generator=g0, device=dev, dtype=torch.float64) * 2.0 ** 67).view(torch.uint8))
        torch.cuda.synchronize()
        eng.key_blob_commit(which)
    blk16 = parse_mux_circuit(open(os.path.join(ROOT, "spf_amd", "data", "mux_multiplier_n16_m16.bincode"), "rb").read())
    rng = np.random.default_rng(2)
    g = FheCircuit(eng)
    b = GraphBuilder(g)
    for _ in range(jobs):
        sel = [b.to_ggsw(g.add_input(ValueKind.GLWE1, rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64))) for _ in range(64)]
        for n in append_uint_multiply(b, sel[:32], sel[32:], lambda x, y: {(16, 16): blk16}[(x, y)]):
            g.add_output(n, ValueKind.GLWE1)
    g.run()
    for _ in range(3):
        t0 = time.perf_counter()
        g.run()
        print(f"run: {(time.perf_counter() - t0) * 1e3:.2f} ms", g.stats())
    g.close()


if __name__ == "__main__":
    main()
