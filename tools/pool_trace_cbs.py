#!/usr/bin/env python3
"""T callers looping KeyswitchL1toL0 -> CircuitBootstrap by handle for a short while: with a -DSPF_POOL_TRACE build of the library
(SPF_HIP_LIBRARY) the per-batch timeline goes to stderr.  usage: pool_trace_cbs.py [T] [seconds]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import bench  # noqa: E402
import spf_amd  # noqa: E402
import tools.driver as drvmod  # noqa: E402
from tools.add32_by_handles import synthetic_engine  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
P = spf_amd.DEFAULT_128
eng = synthetic_engine(P)
drv = drvmod.load()
lib = eng._lib
lwe1 = np.random.default_rng(1).integers(0, 1 << 64, size=P.lwe1_words, dtype=np.uint64)
with bench._pinned_to_quota():
    pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=int(os.environ.get('WAIT_US', '200')))
    ins = [pool.upload(1, lwe1 + np.uint64(t)) for t in range(T)]
    el = C.c_double()
    n = drv.spf_pool_drive_v(pool._h, drvmod.fn(lib, "spf_pool_submit_keyswitch_circuit_bootstrap_v"), drvmod.fn(lib, "spf_pool_wait"),
                             drvmod.fn(lib, "spf_value_release"), T, seconds, drvmod.handles(ins), C.byref(el), None)
    import resource
    ru = resource.getrusage(resource.RUSAGE_SELF)
    print(f"{n} operations in {el.value:.3f} s = {n / el.value:.0f} per second; process CPU so far {ru.ru_utime + ru.ru_stime:.1f} s", file=sys.stderr)
    for v in ins:
        v.release()
    pool.close()
