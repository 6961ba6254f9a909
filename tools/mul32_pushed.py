#!/usr/bin/env python3
"""One 32 x 32 encrypted multiplication (BASELINE config 5's circuit: 127 k CMux + 192 conversions) as ONE gate graph and PUSHED
operation by operation through the pool by ONE thread (pending results as operands, tools/pool_driver.cpp spf_circuit_push):
same words, both timed.  usage: mul32_pushed.py [reps] [bits = 32 | 16] [jobs]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spf_amd  # noqa: E402
import tools.driver as drv  # noqa: E402
from spf_amd import RecordedCircuit, ValueKind  # noqa: E402
from spf_amd.mux_circuits import GraphBuilder, append_uint_multiply, parse_mux_circuit  # noqa: E402
from tools.add32_by_handles import synthetic_engine  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    bits = int(sys.argv[2]) if len(sys.argv) > 2 else 32      # 32: from 16 x 16 blocks; 16: from 8 x 8 blocks
    jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    P = spf_amd.DEFAULT_128
    eng = synthetic_engine(P)
    half = bits // 2
    blk = parse_mux_circuit(open(os.path.join(ROOT, "spf_amd", "data", f"mux_multiplier_n{half}_m{half}.bincode"), "rb").read())
    rng = np.random.default_rng(2)
    rec = RecordedCircuit()
    b = GraphBuilder(rec)
    for _ in range(jobs):
        sel = [b.to_ggsw(rec.add_input(ValueKind.GLWE1, rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64))) for _ in range(2 * bits)]
        for n in append_uint_multiply(b, sel[:bits], sel[bits:], lambda x, y: {(half, half): blk}[(x, y)]):
            rec.add_output(n, ValueKind.GLWE1)
    g, g_outs = rec.lower(eng)
    g.run()
    best_g = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        g.run()
        best_g = min(best_g, time.perf_counter() - t0)
    print(f"graph: {best_g * 1e3:.2f} ms", g.stats(), flush=True)
    pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=1000)
    os.environ.setdefault("SPF_PUSH_TRACE", "1")
    for _ in range(reps + 2):
        t0 = time.perf_counter()
        outs, inner, whole = drv.push_circuit_by_handles(pool, rec)
        c = pool.counters()
        print(f"pushed: {inner * 1e3:.2f} ms inside the pusher ({whole * 1e3:.2f} with upload / download; {(time.perf_counter() - t0) * 1e3:.0f} ms with the "
              f"Python bookkeeping); equal {all(np.array_equal(x, y) for x, y in zip(outs, g_outs))}; launches so far {c['handle_launches']}", flush=True)
    print(pool.value_stats())
    pool.close()
    g.close()


if __name__ == "__main__":
    main()
