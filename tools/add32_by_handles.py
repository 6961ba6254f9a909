#!/usr/bin/env python3
"""K 32-bit encrypted additions (mux_circuits ripple_carry_adder, BASELINE config 3) executed node by node through the pool BY
HANDLES from T native workers (tools/pool_driver.cpp: spf_circuit_drive, the reference's CircuitProcessor in small) beside the
same DAG as ONE gate graph; synthetic keys and ciphertexts (timing is value-independent).
usage: add32_by_handles.py [K] [threads] [max_wait_us] [repeats]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device memory for the synthetic keys only)

import spf_amd  # noqa: E402
import tools.driver as drv  # noqa: E402
from spf_amd.gate_pool import circuit_jobs_as_one_graph  # noqa: E402
from spf_amd.mux_circuits import ripple_carry_adder  # noqa: E402
from spf_amd.sharding import key_blob_tensors, replicate_keys  # noqa: E402


def synthetic_engine(P):
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
    return eng


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    wait_us = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    P = spf_amd.DEFAULT_128
    eng = synthetic_engine(P)
    adder = ripple_carry_adder(32, 32, False)
    cts = np.random.default_rng(3).integers(0, 1 << 64, size=(K, 64, P.glwe_words), dtype=np.uint64)
    rec, _ = circuit_jobs_as_one_graph(eng, adder, cts, record=True)
    g, g_outs = rec.lower(eng)
    g.run()
    best_g = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        g.run()
        best_g = min(best_g, time.perf_counter() - t0)
    print(f"graph: {best_g * 1e3:.3f} ms", g.stats())
    import bench
    pin = bench._pinned_to_quota()
    pin.__enter__()
    pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=wait_us)
    if os.environ.get("SPF_PUSH_ONLY"):      # (for traces: the pushed form alone)
        for _ in range(reps):
            outs, inner, _ = drv.push_circuit_by_handles(pool, rec)
            print(f"pushed: {inner * 1e3:.3f} ms inside the pusher; equal {all(np.array_equal(a, b) for a, b in zip(outs, g_outs))}")
        pool.close()
        g.close()
        return
    outs, _, _ = drv.run_circuit_by_handles(pool, rec, threads=T)
    same = all(np.array_equal(a, b) for a, b in zip(outs, g_outs))
    c0 = pool.counters()
    best = (1e9, 1e9)
    for _ in range(reps):
        _, inner, whole = drv.run_circuit_by_handles(pool, rec, threads=T)
        best = min(best, (whole, inner))
    c1 = pool.counters()
    n_ops = (c1["handle_ops"] - c0["handle_ops"]) // reps
    n_l = (c1["handle_launches"] - c0["handle_launches"]) / reps
    print(f"by handles (K = {K}, {T} threads, max_wait {wait_us} us): {best[0] * 1e3:.3f} ms with upload / download, {best[1] * 1e3:.3f} ms "
          f"inside the driver; {n_ops} operations in {n_l:.0f} launches per run; word-equal to the graph: {same}")
    # the same circuit PUSHED by one thread (pending results as operands, no ticket, no wait until the outputs)
    outs, _, _ = drv.push_circuit_by_handles(pool, rec)
    same = all(np.array_equal(a, b) for a, b in zip(outs, g_outs))
    c0 = pool.counters()
    best = (1e9, 1e9)
    for _ in range(reps):
        _, inner, whole = drv.push_circuit_by_handles(pool, rec)
        best = min(best, (whole, inner))
    c1 = pool.counters()
    n_l = (c1["handle_launches"] - c0["handle_launches"]) / reps
    print(f"pushed by one thread (K = {K}, max_wait {wait_us} us): {best[0] * 1e3:.3f} ms with upload / download, {best[1] * 1e3:.3f} ms "
          f"inside the pusher; {n_l:.0f} launches per run; word-equal to the graph: {same}")
    pool.close()
    g.close()


if __name__ == "__main__":
    main()
