#!/usr/bin/env python3
"""Launch ONE entry point of the library `reps` times at a FIXED batch, device-resident, for rocprofv3: every dispatch of a
kernel in the trace then has the same grid, so per-kernel averages are roofline-grade numbers.
usage: python tools/kernel_bench.py <cmux|keyswitch|cbs|pbs|pbsu|trace|ss> <B> [reps]
  cmux       spf_cmux_dev                 (cmux_kernel; cmux4_kernel for B <= #CU)
  keyswitch  spf_keyswitch_lwe_l1_lwe_l0_dev (ks_digits_kernel + ks_gemm_lds_kernel)
  cbs        spf_circuit_bootstrap_dev    (blind rotation + cbs_trace_kernel + scheme_switch_kernel)
  pbs        spf_circuit_bootstrap_pbs_dev (blind rotation only: blind_rotate4 / 2p2 / 2p by batch size; even rotations)
  pbsu       spf_pbs_univariate_dev       (the plain PBS, log_v = 0: the mixing instantiation of the same kernels)
  trace      spf_mod_switch_trace_and_rotate_dev (cbs_trace_kernel alone, B ciphertexts = 4 B units)
  ss         spf_scheme_switch_dev        (scheme_switch_kernel alone)
Prints one JSON line with the hipEvent time per call."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spf_amd

what, B = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)
eng = spf_amd.Engine(P, device=0)
g = torch.Generator(device=dev).manual_seed(5)
stream = torch.cuda.current_stream().cuda_stream


def rnd_i64(*shape):
    return torch.randint(-(2 ** 63), 2 ** 63 - 1, shape, generator=g, device=dev, dtype=torch.int64)


def key(which, scale):
    ptr, nbytes = eng.key_blob(which)
    from spf_amd.sharding import _DevArray
    t = torch.as_tensor(_DevArray(ptr, nbytes), device=dev)
    if which == 1:
        t.copy_(rnd_i64(nbytes // 8).view(torch.uint8))
    else:
        t.copy_((torch.randn(nbytes // 8, generator=g, device=dev, dtype=torch.float64) * scale).view(torch.uint8))
    torch.cuda.synchronize()  # the blob was filled on torch's stream (spf_key_blob_commit also waits for the device since r05)
    eng.key_blob_commit(which)


if what == "cmux":
    gg = torch.randn((B, P.cbs_ggsw_complex * 2), generator=g, device=dev, dtype=torch.float64) * (2.0 ** 60)
    da, db = rnd_i64(B, P.glwe_words), rnd_i64(B, P.glwe_words)
    dc = torch.empty_like(da)
    call = lambda: eng.cmux_dev(stream, B, gg.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr())
elif what == "keyswitch":
    key(1, 0)
    lwe1 = rnd_i64(B, P.lwe1_words)
    out = torch.empty((B, P.lwe0_words), device=dev, dtype=torch.int64)
    call = lambda: eng.keyswitch_dev(stream, B, lwe1.data_ptr(), out.data_ptr())
elif what in ("trace", "ss"):
    key(2 if what == "trace" else 3, 2.0 ** 67)
    if what == "trace":
        src = rnd_i64(B, P.glwe_words)
        out = torch.empty((B, 4 * P.glwe_words), device=dev, dtype=torch.int64)
        call = lambda: eng.mod_switch_trace_and_rotate_dev(stream, B, src.data_ptr(), out.data_ptr())
    else:
        src = rnd_i64(B, 4 * P.glwe_words)
        out = torch.empty((B, P.cbs_ggsw_complex * 2), device=dev, dtype=torch.float64)
        call = lambda: eng.scheme_switch_dev(stream, B, src.data_ptr(), out.data_ptr())
elif what in ("cbs", "pbs", "pbsu"):
    for which in ((0, 2, 3) if what == "cbs" else (0,)):
        key(which, 2.0 ** 67)
    lwe0 = rnd_i64(B, P.lwe0_words)
    if what == "cbs":
        out = torch.empty((B, P.cbs_ggsw_complex * 2), device=dev, dtype=torch.float64)
        call = lambda: eng.circuit_bootstrap_dev(stream, B, lwe0.data_ptr(), out.data_ptr())
    elif what == "pbsu":
        lut = rnd_i64(P.glwe_words)
        out = torch.empty((B, P.lwe1_words), device=dev, dtype=torch.int64)
        call = lambda: eng.pbs_univariate_dev(stream, B, lwe0.data_ptr(), lut.data_ptr(), 0, out.data_ptr())
    else:
        out = torch.empty((B, P.glwe_words), device=dev, dtype=torch.int64)
        call = lambda: eng.circuit_bootstrap_pbs_dev(stream, B, lwe0.data_ptr(), out.data_ptr())
else:
    raise SystemExit(__doc__)

call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
rec = {"what": what, "B": B, "reps": reps, "ms_per_call": round(e0.elapsed_time(e1) / reps, 4)}
if what in ("pbs", "pbsu", "cmux", "trace", "ss"):
    # checksum of the output words (seeded inputs): bit-equal builds print the same two numbers, so an A/B of library
    # builds (SPF_HIP_LIBRARY) is also a parity check against the build the test-suite verified
    res = (out if what != "cmux" else dc).view(torch.int64).reshape(-1)
    w = torch.arange(res.numel(), device=dev, dtype=torch.int64) * 2654435761 + 12345
    rec["checksum"] = [int(res.sum().item()), int((res * w).sum().item())]
    rec["kernel"] = {"cmux": eng.last_cmux_kernel(), "trace": "cbs_trace_kernel", "ss": "scheme_switch_kernel"}.get(what) or eng.last_blind_rotate_kernel()
print(json.dumps(rec))
