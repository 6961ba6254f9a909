#!/usr/bin/env python3
"""Soak of the per-operation boundary by handles: the reference's 32-bit adder (1 871 operations) executed node by node through
the pool from T native workers (tools/pool_driver.cpp: spf_circuit_drive), PUSHED by one thread without a wait (spf_circuit_push:
pending results as operands), three pushers and a blocking walk at the same time, and two 8 x 8 multiplier blocks pushed — for
`seconds` per thread count, EVERY run's outputs compared word for word with the same DAG as one gate graph; the arena must be
empty at the end.  Synthetic keys and ciphertexts
(the schedule, not the values, is what varies from run to run).
usage: soak_by_handles.py [seconds per thread count] [thread counts, comma separated]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import spf_amd  # noqa: E402
import tools.driver as drv  # noqa: E402
from spf_amd.gate_pool import circuit_jobs_as_one_graph  # noqa: E402
from spf_amd.mux_circuits import ripple_carry_adder  # noqa: E402
from tools.add32_by_handles import synthetic_engine  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,16,64,200").split(",")]
    P = spf_amd.DEFAULT_128
    eng = synthetic_engine(P)
    adder = ripple_carry_adder(32, 32, False)
    rng = np.random.default_rng(11)
    bad = 0
    from concurrent.futures import ThreadPoolExecutor
    from spf_amd.mux_circuits import parse_mux_circuit
    mul8 = None
    try:
        circuit = parse_mux_circuit(open(os.path.join(ROOT, "spf_amd", "data", "mux_multiplier_n8_m8.bincode"), "rb").read())
        mrec, _ = circuit_jobs_as_one_graph(eng, circuit, rng.integers(0, 1 << 64, size=(2, 16, P.glwe_words), dtype=np.uint64), record=True)
        mg, mouts = mrec.lower(eng)
        mg.run()
        mul8 = (mrec, [o.copy() for o in mouts])
        mg.close()
    except FileNotFoundError:
        pass
    for T in counts:
        pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=int(rng.choice([2, 5, 20, 100])))
        t_end = time.time() + seconds
        runs = 0
        while time.time() < t_end:
            cts = rng.integers(0, 1 << 64, size=(1, 64, P.glwe_words), dtype=np.uint64)
            rec, _ = circuit_jobs_as_one_graph(eng, adder, cts, record=True)
            g, g_outs = rec.lower(eng)
            g.run()
            want = [o.copy() for o in g_outs]
            g.close()
            def check(outs, what):
                nonlocal bad, runs
                runs += 1
                if not all(np.array_equal(a, b) for a, b in zip(outs, want)):
                    bad += 1
                    print(f"MISMATCH: {T} threads, run {runs} ({what})", flush=True)

            for _ in range(3):
                check(drv.run_circuit_by_handles(pool, rec, threads=T)[0], "blocking")
            for _ in range(3):      # PUSHED by one thread: pending results as operands, no ticket, no wait until the outputs
                check(drv.push_circuit_by_handles(pool, rec)[0], "pushed")
            with ThreadPoolExecutor(max_workers=4) as ex:   # three pushers and a blocking walk at the same time
                jobs = [ex.submit(drv.push_circuit_by_handles, pool, rec) for _ in range(3)]
                jobs.append(ex.submit(drv.run_circuit_by_handles, pool, rec, T))
                for j in jobs:
                    check(j.result()[0], "concurrent")
            if mul8 is not None:    # the reference's 8 x 8 multiplier block (3 228 CMux in 126 levels), pushed
                outs = drv.push_circuit_by_handles(pool, mul8[0])[0]
                runs += 1
                if not all(np.array_equal(a, b) for a, b in zip(outs, mul8[1])):
                    bad += 1
                    print(f"MISMATCH: multiplier pushed, run {runs}", flush=True)
        c = pool.counters()
        vs = pool.value_stats()
        live, live_bytes, cached = vs["live_values"], vs["live_bytes"], vs["cached_bytes"]
        print(f"{T} threads: {runs} adders, {c['handle_ops']} operations by handle in {c['handle_launches']} launches, mismatches so far {bad}; "
              f"live values {live}, live bytes {live_bytes}, cached {cached >> 20} MiB", flush=True)
        if live or live_bytes:
            bad += 1
        pool.close()
    print("soak:", "FAILED" if bad else "all equal, nothing leaked")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
