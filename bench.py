#!/usr/bin/env python3
"""bench.py — programmable bootstraps per second at DEFAULT_128 on N MI355X GPUs.

One "step" = one pass of the hot path (the circuit-bootstrap PBS: modulus switch, LUT rotate,
637 CMUX steps; sunscreen_tfhe programmable_bootstrapping.rs:342-410 via
circuit_bootstrapping.rs:387-427) over one batch of `--batch` synthetic ciphertexts per GPU,
inputs and keys already resident in HBM.  Workload at N=1: BASELINE.json configs[1]
("Batch of 4096 independent programmable bootstraps, default params, 1xMI355X").

Multi-GPU: one process per GPU (torch.distributed, backend nccl == RCCL).  Bootstraps are
independent, so the batch is sharded with no data-path collective (weak scaling: --batch per
GPU); the only collective is the one-time RCCL broadcast of the evaluation keys from rank 0.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime reads this when it initialises (torch.cuda.is_available() below does that, before the library — whose loader
# sets the same default — is opened): streams that share a hardware queue run their kernels one after the other, and the pool
# keeps several batches resident on the GPU, each on its own stream (profiles/r05_pool.md).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

FLOP_PER_PBS = 263.5e6          # 637 x 413 696 f64 flop
FP64_PEAK_TFLOPS = 78.6         # MI355X dense FP64 (vector == matrix): 256 CU x 128 flop/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md
INT8_MFMA_PEAK_TOPS = 5000.0    # dense int8 MFMA = 2 x the bf16 rate (MI355X_MICROARCH.md, Matrix cores)
# cbs_trace_kernel, per circuit bootstrap: 4 gadget levels x 11 automorphism rounds x
# (6 forward + 2 inverse FFT-1024 at 5*1024*10 flop, 12 pointwise MAD rows at 8*1024, 6 twists at 6*1024, 2 untwists at 8*1024)
TRACE_FLOP_PER_CT = 4 * 11 * (8 * 51200 + 12 * 8192 + 6 * 6144 + 2 * 8192)
# scheme_switch_kernel, per circuit bootstrap: 4 levels x (17 forward + 2 inverse FFT-1024, 30 MAD rows, 17 twists, 2 untwists)
SCHEME_SWITCH_FLOP_PER_CT = 4 * (19 * 51200 + 30 * 8192 + 17 * 6144 + 2 * 8192)


class _stdout_to_stderr:
    """librccl prints a version banner on stdout when a communicator is created; this script's stdout is ONE JSON line.  Routes the
    process's file descriptor 1 to stderr for the duration (C-level prints included)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _host_cpus():
    """CPUs this process can actually run on at once: (min(affinity mask, cgroup CFS quota), CPUs visible, quota in CPUs or
    None).  The GPU box shows 256 CPUs and grants 16 through cpu.max: threads beyond the quota only take turns."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota_cpus = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], txt[1]
            else:
                quota, period = txt[0], open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0]
            if quota not in ("max", "-1"):
                quota_cpus = int(quota) / int(period)
            break
        except (OSError, ValueError, IndexError):
            continue
    usable = visible if quota_cpus is None else max(1, min(visible, int(quota_cpus)))
    return usable, visible, quota_cpus


def _self_launch(gpus: int) -> int:
    """`python3 bench.py --gpus N` from a bare shell (no launcher, WORLD_SIZE unset): start the N ranks as CHILD
    processes through torch.distributed.run and relay their output.  Nothing in this process has touched the GPU
    (no torch import yet), and nothing is exec'ed: the children are fresh interpreters."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed launches first (the first launches of a process run 3-5 %% slower: clocks, key first touch)")
    ap.add_argument("--batch", type=int, default=0,
                    help="ciphertexts per GPU per step (default: 4096 on one GPU = BASELINE configs[1]; 8192 per GPU "
                         "on several = configs[3], 65536 bootstraps over 8 GPUs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline step: skip the gate / CMUX / circuit-bootstrap / 32-bit-add / host-pointer legs "
                         "that the default run appends to the line (a few seconds)")
    ap.add_argument("--with-keyswitch", action="store_true", help="(default on) time the fused gate (keyswitch + PBS)")
    ap.add_argument("--with-cmux", action="store_true", help="(default on) time the batched cbs_radix CMUX kernel")
    ap.add_argument("--with-cbs", action="store_true", help="(default on) time Evaluation::circuit_bootstrap end to end")
    ap.add_argument("--with-add32", type=int, default=-1, metavar="K",
                    help="time K independent 32-bit encrypted additions as ONE gate graph (BASELINE config 3); "
                         "default 1 on a single GPU")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="skip the rocprofv3 --pmc child runs that measure roofline.traffic and the SQ busy fractions of the "
                         "dominant kernel for THIS build on THIS box (three short child processes, about a minute)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo to rehearse "
                    "the N>1 path with several ranks on one GPU)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive all --gpus devices from THIS process through the C ABI's device group (spf_group_*: keys "
                         "replicated inside the library by RCCL, one host thread and stream per device) instead of one "
                         "torch.distributed rank per GPU")
    ap.add_argument("--devices", default="",
                    help="with --single-process: comma-separated HIP ordinals of the group's members (default 0..gpus-1); a "
                         "device may repeat, e.g. 0,0 rehearses a two-member group on a one-GPU box")
    args = ap.parse_args()

    if args.single_process:
        return _single_process_main(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return _self_launch(args.gpus)

    import torch
    import torch.distributed as dist

    import spf_amd
    from spf_amd.sharding import _DevArray, key_blob_tensors, max_over_ranks, replicate_keys

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}: launch with torch.distributed.run "
              f"--nproc-per-node {args.gpus}, or run `python3 bench.py --gpus {args.gpus}` by itself", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path is the only path", file=sys.stderr)
        return 2
    local_dev = local_rank % torch.cuda.device_count()   # == local_rank on a full node
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.batch <= 0:
        args.batch = 4096 if world == 1 else 8192
    extras = not args.no_extras
    args.with_keyswitch = args.with_keyswitch or extras
    args.with_cmux = args.with_cmux or extras
    args.with_cbs = args.with_cbs or extras
    if args.with_add32 < 0:
        args.with_add32 = 1 if (extras and world == 1) else 0
    P = spf_amd.DEFAULT_128
    # only one rank (re)builds the library if it is stale; the others wait for it
    if rank == 0:
        spf_amd.build_library()
    if world > 1:
        dist.barrier(device_ids=[local_dev]) if args.backend == "nccl" else dist.barrier()
    eng = spf_amd.Engine(P, device=local_dev)

    # ---- synthetic evaluation keys: generated on rank 0, RCCL-broadcast into every rank's HBM blob.
    # All four ComputeKey fields (bootstrap, keyswitch, automorphism, scheme switch: crypto/keys.rs:306-318).
    t_keys0 = time.time()
    blobs = key_blob_tensors(eng, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0001)
    if rank == 0:
        # BSK-FFT magnitudes of a real key: DFT of 1024 uniform 64-bit words ~ N(0, (2^63)^2*1024/3)
        bsk = torch.randn(P.bsk_complex * 2, generator=g, device=dev, dtype=torch.float64) * (2.0 ** 67)
        blobs[0].copy_(bsk.view(torch.uint8))
        ksk = torch.randint(-(2 ** 63), 2 ** 63 - 1, (P.ksk_words,), generator=g, device=dev, dtype=torch.int64)
        blobs[1].copy_(ksk.view(torch.uint8))
        del bsk, ksk
        for which in (2, 3):
            n64 = blobs[which].numel() // 8
            blobs[which].copy_((torch.randn(n64, generator=g, device=dev, dtype=torch.float64) * 2.0 ** 67).view(torch.uint8))
    torch.cuda.synchronize()
    t_bcast0 = time.time()
    # RCCL over xGMI, once: 83.5 MB + 62.7 MB + 2.2 MB + 0.5 MB; then every rank commits its replica
    t_bcast, key_bytes = replicate_keys(eng, blobs, dist if world > 1 else None, src=0)
    rccl = None
    if world > 1:
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                "broadcast_s": round(t_bcast, 4), "broadcast_bytes": key_bytes,
                "broadcast_GBs": round(key_bytes / max(t_bcast, 1e-9) / 1e9, 2),
                "note": "first collective of the process group: includes communicator set-up"}

    # ---- synthetic ciphertext batch (uniform torus words; throughput is value-independent)
    B = args.batch
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0100 + rank)
    lwe0 = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.lwe0_words), generator=g, device=dev, dtype=torch.int64)
    glwe_out = torch.empty((B, P.glwe_words), device=dev, dtype=torch.int64)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.circuit_bootstrap_pbs_dev(stream, B, lwe0.data_ptr(), glwe_out.data_ptr())

    def barrier():
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_dev])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = eng.last_kernel_ms("pbs")
    kernel_name = eng.last_blind_rotate_kernel()
    eng.set_timing(False)
    kernel_ms_minmax = [kernel_ms, kernel_ms]
    if world > 1:
        dt = max_over_ranks(dt, dist, device=dev)
        kernel_ms_minmax = [-max_over_ranks(-kernel_ms, dist, device=dev), max_over_ranks(kernel_ms, dist, device=dev)]

    # ---- the other legs of the line.  Each is timed on its own rank between two LOCAL synchronisations (the ranks are
    # independent: weak scaling, no data-path collective) and the MAX over ranks is taken once at the end, so that a
    # leg that fails on one rank is reported in the line instead of leaving the others in a barrier.
    leg_errors = {}

    def leg(name, fn):
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 - reported in the line, never fatal for the headline
            leg_errors[name] = f"{type(e).__name__}: {e}"[:300]
            return None

    def sync():
        torch.cuda.synchronize()

    def restore_headline_output():
        step()   # leave glwe_out holding the plain-PBS result for the parity sample below
        sync()

    def _gate():
        lwe1 = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.lwe1_words), generator=g, device=dev, dtype=torch.int64)
        mid = torch.empty((B, P.lwe0_words), device=dev, dtype=torch.int64)

        def gate_step():
            eng.keyswitch_dev(stream, B, lwe1.data_ptr(), mid.data_ptr())
            eng.circuit_bootstrap_pbs_dev(stream, B, mid.data_ptr(), glwe_out.data_ptr())

        gate_step()
        sync()
        eng.set_timing(True)
        try:
            tg = time.perf_counter()
            for _ in range(args.steps):
                gate_step()
            sync()
            tg = time.perf_counter() - tg
        finally:
            eng.set_timing(False)
            ks_ms, _ = eng.last_kernel_ms("keyswitch")   # (draining: the event pairs are destroyed here)
            eng.last_kernel_ms("pbs")
        ks_ops = 2.0 * B * (P.glwe_size * P.polynomial_degree * P.ks_radix_count) * P.lwe0_words * 8   # int8 MACs x 2 over the 8 byte planes
        ks_tops = ks_ops / (ks_ms * 1e-3) / 1e12 if ks_ms else None
        return {"_seconds": tg, "_units": B * args.steps, "_rate_key": "gates_per_s",
                "keyswitch_kernel_ms": round(ks_ms, 4),
                "keyswitch_int8_mfma_frac": round(ks_tops / INT8_MFMA_PEAK_TOPS, 4) if ks_tops else None,
                "keyswitch_roofline": {"bound": "mfma", "kernel": "ks_digits_kernel + ks_gemm_lds_kernel", "achieved": round(ks_tops, 1) if ks_tops else None,
                                       "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s (int8)",
                                       "frac": round(ks_tops / INT8_MFMA_PEAK_TOPS, 4) if ks_tops else None,
                                       "ops_per_launch": ks_ops, "units_per_launch": B}}

    def _cbs():
        # Evaluation::circuit_bootstrap end to end: PBS -> trace (4 x 11 GLWE keyswitches) -> scheme switch
        ggsw = torch.empty((B, P.cbs_ggsw_complex * 2), device=dev, dtype=torch.float64)
        eng.circuit_bootstrap_dev(stream, B, lwe0.data_ptr(), ggsw.data_ptr())
        sync()
        eng.set_timing(True)
        try:
            tc = time.perf_counter()
            for _ in range(args.steps):
                eng.circuit_bootstrap_dev(stream, B, lwe0.data_ptr(), ggsw.data_ptr())
            sync()
            tc = time.perf_counter() - tc
        finally:
            eng.set_timing(False)
            tr_ms, _ = eng.last_kernel_ms("trace")
            ss_ms, _ = eng.last_kernel_ms("scheme_switch")
            eng.last_kernel_ms("pbs")
        out = {"_seconds": tc, "_units": B * args.steps, "_rate_key": "circuit_bootstraps_per_s",
               "ms_per_batch": round(tc / args.steps * 1e3, 3)}
        if tr_ms:
            tf = TRACE_FLOP_PER_CT * B / (tr_ms * 1e-3) / 1e12
            out["trace_roofline"] = {"bound": "fp64", "kernel": "cbs_trace_kernel", "kernel_ms": round(tr_ms, 4),
                                     "achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(tf / FP64_PEAK_TFLOPS, 4), "flop_per_unit": TRACE_FLOP_PER_CT,
                                     "units_per_launch": B}
        if ss_ms:
            tf = SCHEME_SWITCH_FLOP_PER_CT * B / (ss_ms * 1e-3) / 1e12
            out["scheme_switch_roofline"] = {"bound": "fp64", "kernel": "scheme_switch_kernel", "kernel_ms": round(ss_ms, 4),
                                             "achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                             "frac": round(tf / FP64_PEAK_TFLOPS, 4),
                                             "flop_per_unit": SCHEME_SWITCH_FLOP_PER_CT, "units_per_launch": B}
        return out

    def _pbs_univariate():
        # SURVEY 8(d)'s secondary number and the unit of the reference's own bench (sunscreen_tfhe/benches/ops.rs:86-123):
        # programmable_bootstrap_univariate (programmable_bootstrapping.rs:291-318) = generalized PBS with log_chi = log_v = 0
        # + sample_extract(., 0), identity LUT of one plaintext bit (generate_lut, :129-185).  With log_v = 0 the rotation
        # amounts are odd as often as even: the dispatch takes the instantiation of the blind rotation that keeps the two
        # hand-overs around each rotation gather (nine barriers a step instead of five).
        lut_h = spf_amd.generate_lut([[0, 1]], 1, P)
        d_lut = torch.from_numpy(lut_h.view(np.int64)).to(dev)
        lwe1_out = torch.empty((B, P.lwe1_words), device=dev, dtype=torch.int64)

        def uni_step():
            eng.pbs_univariate_dev(stream, B, lwe0.data_ptr(), d_lut.data_ptr(), 0, lwe1_out.data_ptr())

        uni_step()
        sync()
        eng.set_timing(True)
        try:
            tu = time.perf_counter()
            for _ in range(args.steps):
                uni_step()
            sync()
            tu = time.perf_counter() - tu
        finally:
            eng.set_timing(False)
            u_ms, _ = eng.last_kernel_ms("pbs")
        name = eng.last_blind_rotate_kernel()
        tf = FLOP_PER_PBS * B / (u_ms * 1e-3) / 1e12
        out = {"_seconds": tu, "_units": B * args.steps, "_rate_key": "pbs_per_s", "kernel_ms": round(u_ms, 3),
               "ms_per_step": round(tu / args.steps * 1e3, 3),
               "vs_headline_kernel": round(u_ms / kernel_ms, 4) if kernel_ms else None,
               "lut": "generate_lut(identity, PlaintextBits(1)), shared; output = L1 LWE (sample_extract fused)",
               "roofline": {"bound": "fp64", "kernel": name, "kernel_ms": round(u_ms, 3), "achieved": round(tf, 3),
                            "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_PEAK_TFLOPS, 4),
                            "flop_per_unit": FLOP_PER_PBS, "units_per_launch": B}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # parity on a sample: the oracle's univariate PBS on 64 of the bench ciphertexts (one per host thread)
            import oracle as O
            cnt = min(B, 64)
            bsk_host = blobs[0].cpu().numpy().view(np.complex128)
            _, exp = O.bench_generalized_pbs(lwe0[:cnt].cpu().numpy().view(np.uint64), lut_h, bsk_host, O.DEFAULT_128,
                                             max(1, min(_host_cpus()[0], 64)), 0, 0, extract=True, native=True)
            out["gpu_outputs_bit_equal_on_sample"] = bool(np.array_equal(exp, lwe1_out[:cnt].cpu().numpy().view(np.uint64)))
            out["sample"] = f"{cnt} of the {B} bench ciphertexts against oracle/spf_oracle.c (spfo_pbs_univariate)"
        return out

    def _cmux():
        # KeylessEvaluation::cmux over a batch: every ciphertext brings its own 256 KiB GGSW
        gg = torch.randn((B, P.cbs_ggsw_complex * 2), generator=g, device=dev, dtype=torch.float64) * (2.0 ** 60)
        da = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
        db = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
        dc = torch.empty_like(da)
        eng.cmux_dev(stream, B, gg.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr())
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            eng.cmux_dev(stream, B, gg.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr())
        e1.record()
        sync()
        ms = e0.elapsed_time(e1) / reps
        cm_bytes = B * (P.cbs_ggsw_complex * 16 + 3 * P.glwe_words * 8)
        gbs = cm_bytes / ms / 1e6
        return {"cmux_per_s": round(B / ms * 1e3, 1), "kernel_ms": round(ms, 4),
                "algorithmic_GBs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                "roofline": {"bound": "hbm", "kernel": eng.last_cmux_kernel(),   # the kernel the library actually launched
                             "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                             "bytes_per_unit": P.cbs_ggsw_complex * 16 + 3 * P.glwe_words * 8, "units_per_launch": B}}

    # HBM bytes per launch and the SQ busy fractions of the dominant kernel: measured by this run with rocprofv3 --pmc CHILD processes
    # of this same script — here, BEFORE the legs that create the pools' streams: once this process holds two dozen hardware queues
    # the child's dispatches are time-sliced against them (r06: WRITE_SIZE 854 MB per launch instead of 154, the wave state saved at
    # every switch, and 40.4 ms instead of 38.1 in the child)
    live_early = None
    if rank == 0 and world == 1 and not args.no_live_counters:
        n_cu_early = torch.cuda.get_device_properties(local_dev).multi_processor_count
        live_early = leg("live_counters", lambda: _live_counters(kernel_name, B, kernel_ms, n_cu_early))
    gate = leg("gate", _gate) if args.with_keyswitch else None
    uni = leg("pbs_univariate", _pbs_univariate) if extras else None
    cbs = leg("circuit_bootstrap", _cbs) if args.with_cbs else None
    # (the pool leg runs BEFORE the gate-graph legs: it measures host threads against the GPU, and behind the 511 016-node
    # multiplier graph — 16 GB of arena, a Python heap of a million objects — it read 0.81 / 0.72 / 0.66 of the device-resident
    # rate where the same function alone in a process reads 0.90 / 0.82 / 0.73)
    evpool = leg("evaluation_pool", lambda: _bench_evaluation_pool(eng, P, dev, torch)) if (extras and rank == 0) else None
    # the same callers with device-resident values (r06): pool CMux host-pointer against by handle, circuit bootstrap by handle,
    # the 32-bit adder walked node by node by handles
    byhandle = leg("evaluation_pool_by_handle", lambda: _bench_pool_by_handle(eng, P, dev, torch)) if (extras and rank == 0) else None
    add32h = leg("add32_by_handles", lambda: _bench_add32_by_handles(eng, P)) if (extras and rank == 0 and world == 1) else None
    add32 = None
    if args.with_add32 > 0 and rank == 0:
        add32 = leg("add32", lambda: _bench_add32(eng, P, args.with_add32, dev, g, _DevArray, torch, True))
    mul8 = mul32 = None
    if extras:
        mul8 = leg("mul8_gate_pool", lambda: _bench_mul8_pool(eng, P, rank, world))
        mul32 = leg("mul32_gate_pool", lambda: _bench_mul32_pool(eng, P, rank, world))
    cmux = leg("cmux", _cmux) if args.with_cmux else None
    devgroup = leg("device_group", lambda: _bench_device_group(P, local_dev, lwe0, blobs, torch)) if (extras and rank == 0 and world == 1) else None
    leg("restore", restore_headline_output)

    # one collective for all legs: MAX of the per-rank seconds (inf where a rank failed)
    timed_legs = [("gate", gate), ("circuit_bootstrap", cbs), ("mul8_gate_pool", mul8), ("mul32_gate_pool", mul32),
                  ("pbs_univariate", uni)]
    secs = [(d["_seconds"] if d else float("inf")) for _, d in timed_legs]
    if world > 1:
        t = torch.tensor(secs, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = [float(x) for x in t.tolist()]
    for (name, d), sec in zip(timed_legs, secs):
        if d is None:
            continue
        if sec == float("inf"):
            leg_errors.setdefault(name, "failed on another rank")
            d.clear()
            continue
        d.pop("_seconds")
        units = d.pop("_units") * world
        d[d.pop("_rate_key")] = round(units / sec, 2)
        if "_gates" in d:
            d["gates_per_s"] = round(d.pop("_gates") * world / sec, 1)
        if name.endswith("_pool"):
            d["ms_per_pool_run"] = round(sec * 1e3, 3)
    gate, cbs, mul8, mul32, uni = [(d or None) for _, d in timed_legs]

    total_units = world * B * args.steps
    value = total_units / dt
    ms_per_step = dt / args.steps * 1e3

    # ---- roofline of the dominant kernel (blind_rotate_kernel), per launch
    per_launch_s = kernel_ms * 1e-3 if launches else float("nan")
    achieved_tflops = FLOP_PER_PBS * B / per_launch_s / 1e12
    alg_bytes = P.bsk_complex * 16 + B * (P.lwe0_words * 8 + P.glwe_words * 8)
    # HBM bytes per launch and the SQ busy fractions of the dominant kernel: measured NOW, by this run, with rocprofv3 --pmc
    # child processes of this same script (one per pass, never beside a trace domain; FETCH_SIZE doubled per the gfx950
    # correction, MI355X_MICROARCH.md §HBM).  Falls back to the stored passes of profiles/latest_counters.json (labelled)
    # when rocprofv3 is not on the box or --no-live-counters is given.
    traffic, stored, live = None, None, None
    if rank == 0 and world == 1 and not args.no_live_counters:
        live = live_early
        if live:
            traffic = live.get("hbm_bytes_per_launch")
    tpath = os.path.join(ROOT, "profiles", "latest_counters.json")
    if live is None and os.path.exists(tpath):
        with open(tpath) as f:
            stored = json.load(f)
        if stored.get("kernel") != kernel_name or stored.get("batch") != B:
            stored = {"note": f"stored profile is of {stored.get('kernel')} at batch {stored.get('batch')}: not attached"}
        else:
            traffic = stored.get("hbm_bytes_per_launch")
    roofline = {
        "bound": "fp64", "achieved": round(achieved_tflops, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved_tflops / FP64_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
        "issue_bound_frac": (live or stored or {}).get("valu_busy_frac"),
        "counters": live, "stored_profile": stored, "kernel": kernel_name, "kernel_ms": round(kernel_ms, 3),
        "kernel_ms_min_max_over_ranks": [round(x, 3) for x in kernel_ms_minmax], "launches": launches,
        "flop_per_unit": FLOP_PER_PBS, "units_per_launch": B,
        # what the kernel's own instruction mix allows: of the 2 112 f64 instructions a wave issues per CMUX step 1 700 are
        # v_add_f64 / v_mul_f64 (one flop per lane-slot) and 412 v_fma_f64 (two): 12.0 of the 16 lane-results per cycle the
        # FMA roof counts (profiles/r05_mfma_valu_coexec.md) -> a cap of 0.375 on the nominal flop count at full issue, 0.45
        # counting the executed flops; `frac` stands beside it
        "mix_cap": {"frac_low": 0.375, "frac_high": 0.45, "frac_of_cap": [round(achieved_tflops / FP64_PEAK_TFLOPS / 0.45, 3),
                                                                         round(achieved_tflops / FP64_PEAK_TFLOPS / 0.375, 3)],
                    "note": "ceiling of the transform DAG's add / multiply mix on the FMA roof (1 700 of 2 112 f64 instructions per "
                            "wave-step are one-flop adds or multiplies); r06 microbenchmark: an FMA-folded radix-8 (DAG-II) runs the "
                            "pair's arithmetic 19 % faster (profiles/r06_experiments_blind_rotate.md) — not built: oracle, every "
                            "transform variant and the golden files would have to move together"},
        "note": "the f64 butterflies and MADs run on the VALU (v_fma_f64 / v_add_f64 / v_mul_f64, zero MFMA instructions); "
                "MI355X dense FP64 peak is 78.6 TFLOP/s for VALU and MFMA alike; issue_bound_frac = share of the kernel's "
                "time the VALU is issuing (SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles of the profiled dispatch, GRBM_GUI_ACTIVE / 8); "
                "`counters` = this run's own rocprofv3 --pmc passes (child processes, at most ~150 s in all), `stored_profile` = "
                "the committed passes, used only when no live pass ran",
        "hbm": {"algorithmic_bytes": alg_bytes, "achieved_GBs": round(alg_bytes / per_launch_s / 1e9, 2),
                "peak_GBs": HBM_PEAK_GBS},
    }

    # ---- the host-pointer path (what the Rust shim of INTEGRATION.md calls): numpy arrays in, numpy arrays out,
    # PCIe both ways inside the timed region.  Never `value`.
    pcie = None
    if extras and rank == 0:
        lwe_h = lwe0.cpu().numpy().view(np.uint64)
        # caller-allocated, reused output buffer (what the C ABI's caller holds); the first call also grows
        # the context's staging buffers and touches the pages, outside the timed calls
        out_h = np.zeros((B, P.glwe_words), dtype=np.uint64)
        eng.circuit_bootstrap_pbs(lwe_h, out=out_h)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.circuit_bootstrap_pbs(lwe_h, out=out_h)
        t_h = (time.perf_counter() - t0) / reps
        same = bool(np.array_equal(out_h, glwe_out.cpu().numpy().view(np.uint64)))
        lwe1_h = np.random.default_rng(7).integers(0, 1 << 64, size=(B, P.lwe1_words), dtype=np.uint64)
        eng.gate_bootstrap(lwe1_h, out=out_h)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.gate_bootstrap(lwe1_h, out=out_h)
        t_g = (time.perf_counter() - t0) / reps
        pcie = {"pbs_per_s": round(B / t_h, 1), "ms_per_batch": round(t_h * 1e3, 3),
                "frac_of_device_resident": round((B / t_h) / (B / (dt / args.steps)), 4),
                "gates_per_s": round(B / t_g, 1), "gate_ms_per_batch": round(t_g * 1e3, 3),
                "bytes_in": int(lwe_h.nbytes), "bytes_out": int(out_h.nbytes),
                "note": "spf_circuit_bootstrap_pbs_batch / spf_gate_bootstrap_batch with pageable host arrays: H2D, kernels "
                        "and D2H inside the timed call; outputs leave in slices of one chip round (1024 ciphertexts) under "
                        "the next slice's kernel",
                "same_words_as_device_path": same}
        del out_h

    # ---- CPU baseline: the oracle (a port of the sunscreen_tfhe algorithm), rank 0 at N=1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle as O
        threads, cpus_visible, cfs_quota = _host_cpus()   # one bootstrap per thread on the CPUs that can run at once
        threads = max(1, min(threads, 64))
        OP = O.DEFAULT_128
        bsk_host = blobs[0].cpu().numpy().view(np.complex128)
        # calibrate on one bootstrap, then size the sample for ~cpu_seconds of wall time
        lwe_host = lwe0[: threads * 64].cpu().numpy().view(np.uint64)
        t1, _ = O.bench_cbs_pbs(lwe_host[:1], bsk_host, OP, 1)
        per_thread = max(1, int(args.cpu_seconds / max(t1, 1e-3)))
        count = min(lwe_host.shape[0], threads * per_thread)
        secs, out = O.bench_cbs_pbs(lwe_host[:count], bsk_host, OP, threads)
        gpu_sample = glwe_out[:count].cpu().numpy().view(np.uint64)
        cpu = {"value": round(count / secs, 3), "unit": "PBS/s", "cores": threads, "kind": "port",
               "sample": f"{count} of the {B} bench ciphertexts, one bootstrap per thread on {threads} host threads "
                         f"(oracle/spf_oracle.c, gcc -O3 -march=native, {secs:.1f} s)",
               "cpus_visible": cpus_visible, "cfs_quota_cpus": cfs_quota,
               "single_thread_ms_per_pbs": round(t1 * 1e3, 1),
               "value_over_cores_per_single_thread": round((count / secs) / (threads / max(t1, 1e-9)), 3),
               "gpu_outputs_bit_equal_on_sample": bool(np.array_equal(out, gpu_sample))}

    if rank == 0:
        line = {
            "metric": "programmable bootstraps/sec (CMUX gates/sec) at default 128-bit params",
            "value": round(value, 2), "unit": "PBS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batch of {B} independent circuit-bootstrap PBS per GPU, DEFAULT_128 "
                                   "(n=637, N=2048, k=1, pbs_radix 2x16), keys resident in HBM",
                       "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"batch-sharded x{world}, keys replicated by RCCL broadcast"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "key_broadcast_s": round(t_bcast, 4) if world > 1 else None,
            "key_broadcast_bytes": key_bytes,
            "rccl": rccl,
            "pcie_inclusive": pcie,
            "setup_s": round(t_bcast0 - t_keys0, 2),
        }
        if uni:
            line["pbs_univariate"] = uni
        if gate:
            line["gate"] = gate
        if cmux:
            line["cmux"] = cmux
        if cbs:
            line["circuit_bootstrap"] = cbs
        if evpool:
            line["evaluation_pool"] = evpool
        if byhandle:
            line["evaluation_pool_by_handle"] = byhandle
        if add32h:
            line["add32_by_handles"] = add32h
        if devgroup:
            line["device_group"] = devgroup
        if add32:
            line["add32"] = add32
        if mul8:
            line["mul8_gate_pool"] = mul8
        if mul32:
            line["mul32_gate_pool"] = mul32
        if leg_errors:
            line["leg_errors"] = leg_errors
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    return 0


def _single_process_main(args) -> int:
    """`--gpus N --single-process`: the headline step on N devices driven by ONE host process through spf_group_* — what a
    Rust `Evaluation` owning the whole node does.  Same line shape as the rank-per-GPU form; weak scaling
    (--batch per member); the timed region is K steps enqueued on every member's stream between two synchronisations of
    every device.  Also times the host-pointer group call (the batch cut into ceil(B / G) ranges, PCIe inside)."""
    import torch

    import spf_amd

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path is the only path", file=sys.stderr)
        return 2
    devices = [int(x) for x in args.devices.split(",") if x != ""] or list(range(args.gpus))
    if len(devices) != args.gpus:
        print(f"bench.py: --devices lists {len(devices)} members, --gpus says {args.gpus}", file=sys.stderr)
        return 2
    if max(devices) >= torch.cuda.device_count():
        print(f"bench.py: device {max(devices)} not present ({torch.cuda.device_count()} visible)", file=sys.stderr)
        return 2
    G = len(devices)
    P = spf_amd.DEFAULT_128
    spf_amd.build_library()
    B = args.batch if args.batch > 0 else (4096 if G == 1 else 8192)
    t_keys0 = time.time()
    grp = spf_amd.Group(P, devices=devices)
    members = [grp.member(i) for i in range(G)]
    # synthetic keys generated on member 0's device straight into its key blobs, committed there, then replicated by the
    # library (RCCL broadcast over the distinct devices, device-to-device for a repeated device)
    dev0 = torch.device("cuda", devices[0])
    from spf_amd.sharding import _DevArray
    g = torch.Generator(device=dev0)
    g.manual_seed(0x5EED0001)
    with torch.cuda.device(dev0):
        for which in (0, 1, 2, 3):
            ptr, nbytes = members[0].key_blob(which)
            blob = torch.as_tensor(_DevArray(ptr, nbytes), device=dev0)
            if which == 1:
                src = torch.randint(-(2 ** 63), 2 ** 63 - 1, (nbytes // 8,), generator=g, device=dev0, dtype=torch.int64)
            else:
                src = torch.randn(nbytes // 8, generator=g, device=dev0, dtype=torch.float64) * (2.0 ** 67)
            blob.copy_(src.view(torch.uint8))
            del src
        torch.cuda.synchronize(dev0)
        for which in (0, 1, 2, 3):
            members[0].key_blob_commit(which)
    t_rep0 = time.perf_counter()
    with _stdout_to_stderr():
        grp.replicate_keys()
    t_rep = time.perf_counter() - t_rep0
    rep = grp.replication_stats()

    lwe0, out, streams = [], [], []
    for i, d in enumerate(devices):
        dv = torch.device("cuda", d)
        gi = torch.Generator(device=dv)
        gi.manual_seed(0x5EED0100 + i)
        lwe0.append(torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.lwe0_words), generator=gi, device=dv, dtype=torch.int64))
        out.append(torch.empty((B, P.glwe_words), device=dv, dtype=torch.int64))
        streams.append(torch.cuda.Stream(device=dv))   # one stream per MEMBER (two members on one device do not share one)

    def step():
        for i in range(G):
            members[i].circuit_bootstrap_pbs_dev(streams[i].cuda_stream, B, lwe0[i].data_ptr(), out[i].data_ptr())

    def sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    for _ in range(args.warmup):
        step()
    sync()
    for m in members:
        m.set_timing(True)
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    cpu_headline = time.process_time() - cpu0
    kms = []
    for m in members:
        ms, launches = m.last_kernel_ms("pbs")
        m.set_timing(False)
        kms.append(ms)
    kernel_name = members[0].last_blind_rotate_kernel()
    value = G * B * args.steps / dt
    kernel_ms = max(kms)
    tf = FLOP_PER_PBS * B / (kernel_ms * 1e-3) / 1e12

    # the host-pointer group call: ONE host batch of G x B ciphertexts, cut into G contiguous ranges inside the library
    lwe_h = np.concatenate([x.cpu().numpy().view(np.uint64) for x in lwe0])
    out_h = np.zeros((G * B, P.glwe_words), dtype=np.uint64)
    grp.circuit_bootstrap_pbs(lwe_h, out=out_h)
    reps = 2
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(reps):
        grp.circuit_bootstrap_pbs(lwe_h, out=out_h)
    t_h = (time.perf_counter() - t0) / reps
    cpu_host = (time.process_time() - cpu0) / reps
    same = all(bool(np.array_equal(out_h[i * B:(i + 1) * B], out[i].cpu().numpy().view(np.uint64))) for i in range(G))

    cpu = None
    if G == 1 and not args.no_cpu_baseline:
        import oracle as O
        threads = max(1, min(_host_cpus()[0], 64))
        ptr, nbytes = members[0].key_blob(0)
        bsk_host = torch.as_tensor(_DevArray(ptr, nbytes), device=dev0).cpu().numpy().view(np.complex128)
        cnt = threads
        secs, exp = O.bench_cbs_pbs(lwe_h[:cnt], bsk_host, O.DEFAULT_128, threads)
        cpu = {"value": round(cnt / secs, 3), "unit": "PBS/s", "cores": threads, "kind": "port",
               "sample": f"{cnt} of the bench ciphertexts, one per host thread (oracle/spf_oracle.c)",
               "gpu_outputs_bit_equal_on_sample": bool(np.array_equal(exp, out_h[:cnt]))}

    line = {
        "metric": "programmable bootstraps/sec (CMUX gates/sec) at default 128-bit params",
        "value": round(value, 2), "unit": "PBS/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"batch of {B} independent circuit-bootstrap PBS per GPU, DEFAULT_128 "
                               "(n=637, N=2048, k=1, pbs_radix 2x16), keys resident in HBM",
                   "batch_per_gpu": B, "global_batch": B * G, "devices": devices,
                   "parallelism": f"ONE host process, spf_group of {G} members, batch-sharded, keys replicated in-library"},
        "roofline": {"bound": "fp64", "achieved": round(tf, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / FP64_PEAK_TFLOPS, 4), "traffic": None, "kernel": kernel_name,
                     "kernel_ms": round(kernel_ms, 3), "kernel_ms_min_max_over_members": [round(min(kms), 3), round(max(kms), 3)],
                     "flop_per_unit": FLOP_PER_PBS, "units_per_launch": B,
                     "note": "per member; two members on one device share its CUs (a rehearsal, not a measurement)"
                             if len(set(devices)) < G else "per member"},
        "cpu_baseline": cpu,
        "rccl": {"backend": "rccl (in-library: ncclCommInitAll + in-place ncclBroadcast, single process)"
                            if rep["transport"] == "rccl" else rep["transport"],
                 "world_size": rep["rccl_world_size"] if rep["transport"] == "rccl" else G,
                 "members": G, "distinct_devices": len(set(devices)),
                 "broadcast_s": round(rep["wire_seconds"], 4), "comm_init_s": round(rep["comm_init_seconds"], 4),
                 "replicate_keys_s": round(t_rep, 4), "broadcast_bytes": rep["bytes_per_member"],
                 "broadcast_GBs": round(rep["bytes_per_member"] / max(rep["wire_seconds"], 1e-9) / 1e9, 2)},
        "key_broadcast_s": round(rep["wire_seconds"], 4), "key_broadcast_bytes": rep["bytes_per_member"],
        "pcie_inclusive": {"pbs_per_s": round(G * B / t_h, 1), "ms_per_batch": round(t_h * 1e3, 3),
                           "note": "spf_group_circuit_bootstrap_pbs_batch: one pageable host batch of G x B ciphertexts cut into G "
                                   "contiguous ranges, each member's H2D / kernels / sliced D2H on its own thread and stream",
                           "same_words_as_device_path": same,
                           "host_cpu_s_per_batch": round(cpu_host, 3), "host_cpu_s_per_member_per_batch": round(cpu_host / G, 4)},
        "host_cpu": {"headline_cpu_s_per_step": round(cpu_headline / args.steps, 5),
                     "note": "process CPU time (all threads): the device-pointer headline only enqueues; the host-pointer call "
                             "stages every member's range through its own thread"},
        "setup_s": round(time.time() - t_keys0, 2),
    }
    if len(set(devices)) < G:
        line["rehearsal"] = ("members share a device: a rehearsal of the host path (replication, split, dealing, reassembly, host "
                             "CPU per member), NOT a scaling figure")
    if not args.no_extras:
        for name, fn in (("mul8_gate_pool", lambda: _group_gate_pool(grp, P, G, "mul8", 8)),
                         ("add32", lambda: _group_gate_pool(grp, P, G, "add32", 4)),
                         ("mul32_gate_pool", lambda: _group_gate_pool(grp, P, G, "mul32", 1)),
                         ("evaluation_pool_by_handle", lambda: _group_pool_by_handle(grp, P, G))):
            t_leg = time.time()
            try:
                line[name] = fn()
                line[name]["leg_s"] = round(time.time() - t_leg, 1)
            except Exception as e:   # a failing leg is recorded, never hidden
                line.setdefault("leg_errors", {})[name] = f"{type(e).__name__}: {e}"
    print(json.dumps(line))
    grp.close()
    return 0


def _group_gate_pool(grp, P, G, what, per_member):
    """BASELINE config 5's shape from ONE process: `per_member` x G independent circuits as jobs of the device group
    (spf_group_graph_create), dealt by cost and run side by side by spf_group_run_graphs; the same jobs on member 0 alone for
    comparison.  Synthetic ciphertexts (timing is value-independent; correctness: tests/test_gpu_group.py)."""
    import spf_amd
    from spf_amd import ValueKind
    from spf_amd.gate_pool import circuit_jobs_as_one_graph
    from spf_amd.mux_circuits import GraphBuilder, append_uint_multiply, parse_mux_circuit, ripple_carry_adder
    rng = np.random.default_rng(0x6A0B)
    n_jobs = per_member * G

    def make(engine):
        if what == "mul8":
            c = parse_mux_circuit(open(os.path.join(DATA_DIR, "mux_multiplier_n8_m8.bincode"), "rb").read())
            return circuit_jobs_as_one_graph(engine, c, rng.integers(0, 1 << 64, size=(1, 16, P.glwe_words), dtype=np.uint64))[0]
        if what == "add32":
            return circuit_jobs_as_one_graph(engine, ripple_carry_adder(32, 32, False),
                                             rng.integers(0, 1 << 64, size=(1, 64, P.glwe_words), dtype=np.uint64))[0]
        blk16 = parse_mux_circuit(open(os.path.join(DATA_DIR, "mux_multiplier_n16_m16.bincode"), "rb").read())
        g = spf_amd.FheCircuit(engine)
        b = GraphBuilder(g)
        sel = [b.to_ggsw(g.add_input(ValueKind.GLWE1, rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64))) for _ in range(64)]
        for n in append_uint_multiply(b, sel[:32], sel[32:], lambda x, y: {(16, 16): blk16}[(x, y)]):
            g.add_output(n, ValueKind.GLWE1)
        return g

    t_build = time.perf_counter()
    jobs = [make(grp) for _ in range(n_jobs)]
    t_build = time.perf_counter() - t_build
    grp.run_graphs(jobs)            # deals, merges, plans, warms up
    reps = 2
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(reps):
        grp.run_graphs(jobs)
    dt = (time.perf_counter() - t0) / reps
    cpu = (time.process_time() - cpu0) / reps
    placement = [0] * G
    for j in jobs:
        placement[j.member()] += 1
    st = jobs[0].stats()
    out = {"jobs": n_jobs, "per_member": per_member, "members": G, "placement": placement,
           "ms_per_pool_run": round(dt * 1e3, 3), "jobs_per_s": round(n_jobs / dt, 2),
           "host_cpu_s_per_pool_run": round(cpu, 4), "graph_build_s": round(t_build, 2),
           "nodes_per_job": st["nodes"], "levels": st["levels"], "launches_per_member": st["launches"]}
    for j in jobs:
        j.close()
    return out


def _group_pool_by_handle(grp, P, G, threads_per_member=64, seconds=2.0):
    """The per-operation drop-in over the group from one process: threads_per_member x G native callers loop
    KeyswitchL1toL0 -> CircuitBootstrap by handle; caller t's input lives on member t mod G (a value stays on its member, an
    operation runs where its operands live)."""
    import ctypes as C
    import spf_amd
    import tools.driver as drvmod
    drv = drvmod.load()
    lib = grp._raw
    T = threads_per_member * G
    lwe1 = np.random.default_rng(0x9003).integers(0, 1 << 64, size=P.lwe1_words, dtype=np.uint64)
    with _pinned_to_quota():
        pool = spf_amd.Pool(grp, max_batch=4096, max_wait_us=200)
        try:
            ins = [pool.upload(1, lwe1 + np.uint64(t), member=t % G) for t in range(T)]
            el = C.c_double()
            argv = (pool._h, drvmod.fn(lib, "spf_pool_submit_keyswitch_circuit_bootstrap_v"), drvmod.fn(lib, "spf_pool_wait"),
                    drvmod.fn(lib, "spf_value_release"), T)
            drv.spf_pool_drive_v(*argv, 0.5, drvmod.handles(ins), C.byref(el), None)
            c0 = pool.counters()
            cpu0 = time.process_time()
            n = drv.spf_pool_drive_v(*argv, seconds, drvmod.handles(ins), C.byref(el), None)
            cpu = time.process_time() - cpu0
            c1 = pool.counters()
            for v in ins:
                v.release()
        finally:
            pool.close()
    if n < 0:
        raise RuntimeError("pool driver: a circuit bootstrap by handle failed")
    return {"threads": T, "threads_per_member": threads_per_member, "circuit_bootstraps_per_s": round(n / el.value, 1),
            "achieved_batch": round((c1["handle_ops"] - c0["handle_ops"]) / max(1, c1["handle_launches"] - c0["handle_launches"]), 1),
            "host_cpu_s_per_s": round(cpu / el.value, 2)}


DATA_DIR = os.path.join(ROOT, "spf_amd", "data")


def _live_counters(kernel_name, B, kernel_ms, n_cu, budget_s=150.0):
    """rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 --no-extras` (the headline step only) as CHILD processes,
    one pass per process: HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KB counters) and the SQ busy / wait fractions
    of the blind-rotation kernel.  The program behind `--` is python3 itself (no env / shell hop).  Busy fractions divide the
    SQ counters by the cycles of the PROFILED dispatch they were counted in (GRBM_GUI_ACTIVE / 8 XCDs, collected in the same
    pass; MI355X_MICROARCH.md, DVFS give-back) — not by an assumed clock — and the SIMD / CU counts come from the device.
    The whole thing is bounded by `budget_s` of wall time: a pass that would not fit is skipped and named in `skipped`."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    # never nest: under a profiler this process carries its preloaded tool library, a child launcher would inherit it,
    # initialise the GPU before it execs python3, and that exec is refused on this pool (and pointless anyway)
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or \
            any(k.startswith(("ROCPROF", "ROCPROFILER_")) for k in os.environ):
        raise RuntimeError("already running under a profiler: live counter passes skipped")
    passes = [["SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU",
               "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"],
              ["FETCH_SIZE"], ["WRITE_SIZE"]]
    base = kernel_name.split("<")[0]
    acc = collections.defaultdict(list)
    env = dict(os.environ, TMPDIR="/tmp")
    t_start, longest, skipped, child_ms = time.time(), 0.0, [], None
    with tempfile.TemporaryDirectory(prefix="spf_pmc_", dir="/tmp") as tmp:
        for i, counters in enumerate(passes):
            left = budget_s - (time.time() - t_start)
            if left < max(20.0, 1.3 * longest):
                skipped.append(counters[0])
                continue
            out = os.path.join(tmp, f"p{i}")
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "1", "--batch", str(B), "--no-extras", "--no-cpu-baseline", "--no-live-counters"]
            t0 = time.time()
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=left, cwd="/tmp")
            except subprocess.TimeoutExpired:
                skipped.append(counters[0] + " (timed out)")
                break
            longest = max(longest, time.time() - t0)
            if r.returncode != 0:
                skipped.append(counters[0] + f" (rc {r.returncode}: {r.stderr[-160:]})")
                break
            if i == 0:   # the SQ pass: its own hipEvent kernel time, i.e. the time of the dispatches the counters belong to
                for ln in r.stdout.splitlines():
                    if ln.startswith("{"):
                        child_ms = json.loads(ln)["roofline"]["kernel_ms"]
            for f in glob.glob(os.path.join(out, "*", "*_counter_collection.csv")):
                for row in csv.DictReader(open(f)):
                    if base in row["Kernel_Name"]:
                        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items() if v}
    if not c:
        raise RuntimeError("no counter rows for " + base + (": " + "; ".join(skipped) if skipped else ""))
    out = {"source": "rocprofv3 --pmc child passes of this run (SQ_* + GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE), per dispatch of " + base,
           "wall_s": round(time.time() - t_start, 1)}
    if skipped:
        out["skipped"] = skipped
    if "FETCH_SIZE" in c:
        out["FETCH_SIZE_KB"], out["WRITE_SIZE_KB"] = c["FETCH_SIZE"], c.get("WRITE_SIZE")
        if "WRITE_SIZE" in c:
            out["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0            # cycles of the profiled dispatch (the counter sums the 8 XCDs)
        n_simd = 4 * n_cu
        out["profiled_kernel_ms"] = child_ms
        out["profiled_clock_GHz"] = round(cyc / (child_ms * 1e-3) / 1e9, 3) if child_ms else None
        out["valu_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / n_simd / cyc, 4)   # SQ_ACTIVE_* count quad-cycles
        out["lds_busy_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / n_cu / cyc, 4)
        out["wait_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        out["issue_stall_frac"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
        out["SQ_INSTS_VALU"] = c["SQ_INSTS_VALU"]
        out["SQ_LDS_BANK_CONFLICT"] = c["SQ_LDS_BANK_CONFLICT"]
    return out


def _bench_device_group(P, device, lwe0, blobs, torch, batch=2048):
    """The multi-GPU boundary on the box at hand (include/spf_hip.h `spf_group_*`): ONE host process, a group of two
    members on this one GPU — keys uploaded once from the host and replicated INSIDE the library (a one-rank RCCL communicator +
    in-place broadcast, device-to-device copy for the second member), one host batch cut into two contiguous ranges, each on its
    member's thread and stream, results reassembled.  A rehearsal of the mechanism, not a scaling measurement: both members
    share this GPU's CUs.  Reports the replication record and the host-pointer group call against the single context."""
    import spf_amd
    keys_host = [b.cpu().numpy() for b in blobs]
    grp = spf_amd.Group(P, devices=[device, device])
    try:
        t0 = time.perf_counter()
        with _stdout_to_stderr():
            grp.load_bootstrap_key(keys_host[0].view(np.complex128))
            grp.load_keyswitch_key(keys_host[1].view(np.uint64))
            grp.load_automorphism_key(keys_host[2].view(np.complex128))
            grp.load_scheme_switch_key(keys_host[3].view(np.complex128))
        t_load = time.perf_counter() - t0
        rep = grp.replication_stats()
        B = min(batch, lwe0.shape[0])
        lwe_h = lwe0[:B].cpu().numpy().view(np.uint64)
        out_g = np.zeros((B, P.glwe_words), dtype=np.uint64)
        grp.circuit_bootstrap_pbs(lwe_h, out=out_g)
        t0 = time.perf_counter()
        grp.circuit_bootstrap_pbs(lwe_h, out=out_g)
        t_g = time.perf_counter() - t0
        one = grp.member(0)
        out_1 = np.zeros_like(out_g)
        one.circuit_bootstrap_pbs(lwe_h, out=out_1)
        t0 = time.perf_counter()
        one.circuit_bootstrap_pbs(lwe_h, out=out_1)
        t_1 = time.perf_counter() - t0
        # a member drained and re-admitted: the split follows the rotation, the words do not change
        grp.set_member_enabled(1, False)
        out_d = np.zeros_like(out_g)
        grp.circuit_bootstrap_pbs(lwe_h[:64], out=out_d[:64])
        grp.set_member_enabled(1, True)
        return {"members": 2, "devices": [device, device], "transport": rep["transport"], "rccl_world_size": rep["rccl_world_size"],
                "key_bytes_replicated_per_member": rep["bytes_per_member"], "replication_wire_s": round(rep["wire_seconds"], 4),
                "rccl_comm_init_s": round(rep["comm_init_seconds"], 3), "load_and_replicate_s": round(t_load, 3),
                "batch": B, "group_ms_per_batch": round(t_g * 1e3, 3), "single_context_ms_per_batch": round(t_1 * 1e3, 3),
                "same_words_as_single_context": bool(np.array_equal(out_g, out_1) and np.array_equal(out_d[:64], out_1[:64])),
                "note": "two members on ONE GPU share its CUs: a rehearsal of replication / split / reassembly, not a scaling figure"}
    finally:
        grp.close()


def _bench_evaluation_pool(eng, P, dev, torch, thread_counts=(64, 256, 1024), seconds=2.5):
    """The drop-in scenario itself (VERDICT r3 task 5): T host threads, each calling `FheOp::KeyswitchL1toL0 ->
    FheOp::CircuitBootstrap` on ONE ciphertext synchronously, again and again — what the rayon workers of
    `CircuitProcessor::execute_task` do (parasol_runtime/src/circuit_processor/mod.rs:192-253) — through spf_pool_*.
    The callers are native threads (tools/pool_driver.cpp: Python threads would serialise on the interpreter lock); they use
    the public C ABI only; they are confined to the box's CPU share (the GPU box grants 16 CPUs through a CFS quota while
    `nproc` shows 256: a thousand threads waking across 256 CPUs spend the whole quota in scheduler work — 19 s of kernel
    time per 1.8 s of wall time, 12 of 17 periods throttled, tools/pool_probe.py — so the leg pins itself to the quota's
    CPUs for its duration).  Reported per T: circuit bootstraps per second, the batch size the pool achieved
    (spf_pool_stats), and the ratio to the device-resident rate of the same two kernels at a batch of T (inputs and
    outputs in HBM, no host traffic): 16 KiB in and 256 KiB out per operation cross PCIe here."""
    import ctypes as C
    import subprocess
    import spf_amd
    drv_path = os.path.join(ROOT, "tools", "bin", "libpool_driver.so")
    src = os.path.join(ROOT, "tools", "pool_driver.cpp")
    if not os.path.exists(drv_path) or os.path.getmtime(drv_path) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(drv_path), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "include"),
                        "-o", drv_path, src], check=True)
    drv = C.CDLL(drv_path)
    drv.spf_pool_drive.restype = C.c_long
    drv.spf_pool_drive.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_size_t,
                                   C.POINTER(C.c_double), C.POINTER(C.c_double)]
    cpus_before = os.sched_getaffinity(0)
    n_quota = _host_cpus()[0]
    os.sched_setaffinity(0, set(sorted(cpus_before)[:n_quota]))   # the driver's threads inherit it
    out_cpus = n_quota
    lib = eng._lib
    submit = C.cast(lib.spf_pool_submit_keyswitch_circuit_bootstrap, C.c_void_p)
    wait = C.cast(lib.spf_pool_wait, C.c_void_p)
    lwe1 = np.random.default_rng(0x9001).integers(0, 1 << 64, size=P.lwe1_words, dtype=np.uint64)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"op": "spf_pool_submit_keyswitch_circuit_bootstrap + spf_pool_wait, one ciphertext per call",
           "host_cpus": out_cpus, "runs": []}
    try:
        _pool_runs(out, thread_counts, seconds, eng, P, dev, torch, drv, submit, wait, lwe1, stream, spf_amd, C)
    finally:
        os.sched_setaffinity(0, cpus_before)
    return out


def _pool_runs(out, thread_counts, seconds, eng, P, dev, torch, drv, submit, wait, lwe1, stream, spf_amd, C):
    for T in thread_counts:
        # device-resident reference at a batch of T: keyswitch + whole circuit bootstrap, everything in HBM
        d_in = torch.randint(-(2 ** 63), 2 ** 63 - 1, (T, P.lwe1_words), device=dev, dtype=torch.int64)
        d_mid = torch.empty((T, P.lwe0_words), device=dev, dtype=torch.int64)
        d_out = torch.empty((T, P.cbs_ggsw_complex * 2), device=dev, dtype=torch.float64)

        def dev_step():
            eng.keyswitch_dev(stream, T, d_in.data_ptr(), d_mid.data_ptr())
            eng.circuit_bootstrap_dev(stream, T, d_mid.data_ptr(), d_out.data_ptr())

        dev_step()
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            dev_step()
        torch.cuda.synchronize()
        dev_rate = T * reps / (time.perf_counter() - t0)
        del d_in, d_mid, d_out
        pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=200)
        try:
            el, ck = C.c_double(), C.c_double()
            drv.spf_pool_drive(pool._h, submit, wait, T, 0.5, lwe1.ctypes.data, P.lwe1_words, P.cbs_ggsw_complex * 2,
                               C.byref(el), C.byref(ck))   # warm-up: staging buffers pinned, shapes settled
            ops0, launches0 = pool.stats()
            c0 = pool.counters()
            n = drv.spf_pool_drive(pool._h, submit, wait, T, seconds, lwe1.ctypes.data, P.lwe1_words, P.cbs_ggsw_complex * 2,
                                   C.byref(el), C.byref(ck))
            ops1, launches1 = pool.stats()
            c1 = pool.counters()
        finally:
            pool.close()
        if n < 0:
            raise RuntimeError(f"pool driver: an operation failed at T = {T}")
        rate = n / el.value
        out["runs"].append({"threads": T, "circuit_bootstraps_per_s": round(rate, 1), "operations": int(n),
                            "achieved_batch": round((ops1 - ops0) / max(1, launches1 - launches0), 1),
                            "launches_by_shape": {k: c1["bootstrap_launches_by_shape"][k] - c0["bootstrap_launches_by_shape"][k]
                                                  for k in c1["bootstrap_launches_by_shape"]},
                            "device_resident_rate_at_batch_T": round(dev_rate, 1),
                            "frac_of_device_resident": round(rate / dev_rate, 4)})


def _pinned_to_quota():
    """context: the process (and the native threads it starts) confined to the CPUs its CFS quota grants (see
    _bench_evaluation_pool)"""
    import contextlib

    @contextlib.contextmanager
    def scope():
        before = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(sorted(before)[:_host_cpus()[0]]))
        try:
            yield
        finally:
            os.sched_setaffinity(0, before)
    return scope()


def _bench_pool_by_handle(eng, P, dev, torch, thread_counts=(64, 256, 1024), seconds=2.0, cmux_cases=((64, 1), (256, 1), (16, 64)),
                          cbs_wait_us=200):
    """The drop-in scenario BY HANDLE (r06, include/spf_hip.h "device-resident values"): the same T native callers, one operation
    per call, operands and results device-resident values — nothing crosses PCIe per call.
      cmux            T callers loop one CMux gate each (the reference's `FheOp::CMux` task, circuit_processor/mod.rs:391-421):
                      host-pointer submits (256 KiB + 2 x 32 KiB up, 32 KiB down per gate) against submits by handle — with
                      one ticket open per thread (submit, wait: the synchronous call of the reference, bounded by a thread
                      sleep and wake-up per gate) and with 64 tickets open per thread (the pool's asynchronous use)
      circuit_bootstrap  T callers loop KeyswitchL1toL0 -> CircuitBootstrap by handle (the GGSW stays in HBM), against the
                      device-resident rate of the same kernels at a batch of T"""
    import ctypes as C
    import spf_amd
    import tools.driver as drvmod
    drv = drvmod.load()
    lib = eng._lib
    wait = drvmod.fn(lib, "spf_pool_wait")
    release = drvmod.fn(lib, "spf_value_release")
    rng = np.random.default_rng(0x9002)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"host_cpus": _host_cpus()[0], "cmux": [], "circuit_bootstrap": []}
    sel = (rng.standard_normal(P.cbs_ggsw_complex * 2) * 2.0 ** 58)
    a = rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64)
    b = rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64)
    lwe1 = rng.integers(0, 1 << 64, size=P.lwe1_words, dtype=np.uint64)
    with _pinned_to_quota():
        for T, window in cmux_cases:
            row = {"threads": T, "tickets_open_per_thread": window}
            pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=20)
            try:
                el = C.c_double()
                args = (pool._h, drvmod.fn(lib, "spf_pool_submit_cmux"), wait, T, window)
                drv.spf_pool_drive_cmux(*args, 0.3, sel.ctypes.data, sel.size, a.ctypes.data, b.ctypes.data, a.size, C.byref(el))
                n = drv.spf_pool_drive_cmux(*args, seconds, sel.ctypes.data, sel.size, a.ctypes.data, b.ctypes.data, a.size, C.byref(el))
                if n < 0:
                    raise RuntimeError("pool driver: a host-pointer CMux failed")
                row["host_pointer_gates_per_s"] = round(n / el.value, 1)
                vs = pool.upload_batch(3, np.tile(sel.view(np.complex128), (T, 1)))
                va = pool.upload_batch(2, np.tile(a, (T, 1)))
                vb = pool.upload_batch(2, np.tile(b, (T, 1)))
                hs, ha, hb = drvmod.handles(vs), drvmod.handles(va), drvmod.handles(vb)
                argv = (pool._h, drvmod.fn(lib, "spf_pool_submit_cmux_v"), wait, release, T, window)
                drv.spf_pool_drive_cmux_v(*argv, 0.3, hs, ha, hb, C.byref(el))
                c1 = pool.counters()
                n = drv.spf_pool_drive_cmux_v(*argv, seconds, hs, ha, hb, C.byref(el))
                c2 = pool.counters()
                if n < 0:
                    raise RuntimeError("pool driver: a CMux by handle failed")
                row["by_handle_gates_per_s"] = round(n / el.value, 1)
                row["by_handle_achieved_batch"] = round((c2["handle_ops"] - c1["handle_ops"]) / max(1, c2["handle_launches"] - c1["handle_launches"]), 1)
                row["by_handle_over_host_pointer"] = round(row["by_handle_gates_per_s"] / row["host_pointer_gates_per_s"], 2)
                for v in vs + va + vb:
                    v.release()
            finally:
                pool.close()
            out["cmux"].append(row)
        # ... and PUSHED: bursts of gates without tickets, two bursts in flight per thread, the values of a burst waited for before
        # its slots are reused (tools/pool_driver.cpp spf_pool_push_cmux_v)
        for T, burst in (((1, 2048), (4, 1024)) if cmux_cases else ()):
            pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=20)
            try:
                vs = pool.upload_batch(3, np.tile(sel.view(np.complex128), (T, 1)))
                va = pool.upload_batch(2, np.tile(a, (T, 1)))
                vb = pool.upload_batch(2, np.tile(b, (T, 1)))
                hs, ha, hb = drvmod.handles(vs), drvmod.handles(va), drvmod.handles(vb)
                el = C.c_double()
                argp = (pool._h, drvmod.fn(lib, "spf_pool_submit_cmux_v"), drvmod.fn(lib, "spf_value_wait"), release, T, burst)
                drv.spf_pool_push_cmux_v(*argp, 0.3, hs, ha, hb, C.byref(el))
                c1 = pool.counters()
                n = drv.spf_pool_push_cmux_v(*argp, seconds, hs, ha, hb, C.byref(el))
                c2 = pool.counters()
                if n < 0:
                    raise RuntimeError("pool driver: a pushed CMux failed")
                for v in vs + va + vb:
                    v.release()
            finally:
                pool.close()
            host_rate = out["cmux"][0]["host_pointer_gates_per_s"]
            out["cmux"].append({"threads": T, "pushed_burst": burst, "by_handle_gates_per_s": round(n / el.value, 1),
                                "by_handle_achieved_batch": round((c2["handle_ops"] - c1["handle_ops"]) / max(1, c2["handle_launches"] - c1["handle_launches"]), 1),
                                "by_handle_over_host_pointer": round(n / el.value / host_rate, 2)})
        for T in thread_counts:
            d_in = torch.randint(-(2 ** 63), 2 ** 63 - 1, (T, P.lwe1_words), device=dev, dtype=torch.int64)
            d_mid = torch.empty((T, P.lwe0_words), device=dev, dtype=torch.int64)
            d_out = torch.empty((T, P.cbs_ggsw_complex * 2), device=dev, dtype=torch.float64)

            def dev_step():
                eng.keyswitch_dev(stream, T, d_in.data_ptr(), d_mid.data_ptr())
                eng.circuit_bootstrap_dev(stream, T, d_mid.data_ptr(), d_out.data_ptr())

            dev_step()
            torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                dev_step()
            torch.cuda.synchronize()
            dev_rate = T * reps / (time.perf_counter() - t0)
            del d_in, d_mid, d_out
            pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=cbs_wait_us)
            try:
                ins = [pool.upload(1, lwe1 + np.uint64(t)) for t in range(T)]
                hin = drvmod.handles(ins)
                el = C.c_double()
                argv = (pool._h, drvmod.fn(lib, "spf_pool_submit_keyswitch_circuit_bootstrap_v"), wait, release, T)
                drv.spf_pool_drive_v(*argv, 0.5, hin, C.byref(el), None)
                c0 = pool.counters()
                n = drv.spf_pool_drive_v(*argv, seconds, hin, C.byref(el), None)
                c1 = pool.counters()
                for v in ins:
                    v.release()
            finally:
                pool.close()
            if n < 0:
                raise RuntimeError(f"pool driver: a circuit bootstrap by handle failed at T = {T}")
            rate = n / el.value
            shapes = {k: c1["bootstrap_launches_by_shape"][k] - c0["bootstrap_launches_by_shape"][k] for k in c1["bootstrap_launches_by_shape"]}
            out["circuit_bootstrap"].append({"threads": T, "circuit_bootstraps_per_s": round(rate, 1), "operations": int(n),
                                             "achieved_batch": round((c1["handle_ops"] - c0["handle_ops"]) / max(1, c1["handle_launches"] - c0["handle_launches"]), 1),
                                             "launches_by_shape": shapes,
                                             "device_resident_rate_at_batch_T": round(dev_rate, 1),
                                             "frac_of_device_resident": round(rate / dev_rate, 4)})
    return out


def _bench_add32_by_handles(eng, P, threads=64, reps=5):
    """BASELINE config 3 the way the reference runs it: the adder's DAG walked node by node (one task per `FheOp`, each ONE
    spf_pool_submit_op_v + spf_pool_wait from a pool of native workers: tools/pool_driver.cpp spf_circuit_drive =
    circuit_processor/mod.rs:130-253 in small), values device-resident, against the same DAG as ONE gate graph."""
    import spf_amd
    import tools.driver as drvmod
    from spf_amd.gate_pool import circuit_jobs_as_one_graph
    from spf_amd.mux_circuits import ripple_carry_adder
    adder = ripple_carry_adder(32, 32, False)
    cts = np.random.default_rng(3).integers(0, 1 << 64, size=(1, 64, P.glwe_words), dtype=np.uint64)
    rec, _ = circuit_jobs_as_one_graph(eng, adder, cts, record=True)
    g, g_outs = rec.lower(eng)
    g.run()
    best_g = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        g.run()
        best_g = min(best_g, time.perf_counter() - t0)
    with _pinned_to_quota():
        pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=5)
        try:
            outs, _, _ = drvmod.run_circuit_by_handles(pool, rec, threads=threads)
            same = all(np.array_equal(x, y) for x, y in zip(outs, g_outs))
            c0 = pool.counters()
            best = (1e9, 1e9)
            for _ in range(reps):
                _, inner, whole = drvmod.run_circuit_by_handles(pool, rec, threads=threads)
                best = min(best, (whole, inner))
            c1 = pool.counters()
        finally:
            pool.close()
        # the same DAG PUSHED by ONE thread: every operation one spf_pool_submit_op_v without a ticket, operands that are still
        # pending, no wait until the outputs (spf_value_wait) — include/spf_hip.h "Deferred operands", tools/pool_driver.cpp
        # spf_circuit_push.  (A long quiet time: the wait for the outputs launches what was pushed.)
        pool = spf_amd.Pool(eng, max_batch=4096, max_wait_us=1000)
        try:
            for _ in range(3):      # (the sets' pointer tables and the batch sizes grow to the circuit's levels)
                outs, _, _ = drvmod.push_circuit_by_handles(pool, rec)
            same_p = all(np.array_equal(x, y) for x, y in zip(outs, g_outs))
            p0 = pool.counters()
            best_p = (1e9, 1e9)
            for _ in range(reps):
                _, inner, whole = drvmod.push_circuit_by_handles(pool, rec)
                best_p = min(best_p, (whole, inner))
            p1 = pool.counters()
        finally:
            pool.close()
    g.close()
    return {"threads": threads, "operations": (c1["handle_ops"] - c0["handle_ops"]) // reps,
            "launches": round((c1["handle_launches"] - c0["handle_launches"]) / reps, 1),
            "ms_per_add_by_handles": round(best[0] * 1e3, 3), "ms_inside_the_driver": round(best[1] * 1e3, 3),
            "ms_per_add_as_one_graph": round(best_g * 1e3, 3), "by_handles_over_graph": round(best[0] / best_g, 2),
            "word_equal_to_the_graph": bool(same),
            "pushed": {"threads": 1, "launches": round((p1["handle_launches"] - p0["handle_launches"]) / reps, 1),
                       "ms_per_add": round(best_p[0] * 1e3, 3), "ms_inside_the_pusher": round(best_p[1] * 1e3, 3),
                       "over_graph": round(best_p[0] / best_g, 2), "word_equal_to_the_graph": bool(same_p)}}


def _bench_mul8_pool(eng, P, rank, world, per_gpu=8):
    """BASELINE config 5's shape: a pool of independent multiplications through the reference's 8 x 8 multiplier
    block (mux_circuits `unsigned_multiplier(8, 8)`: 3 228 CMUX in 126 levels + 16 bit conversions each), `per_gpu`
    jobs per GPU (weak scaling), dealt to the ranks by spf_amd.gate_pool, each rank's jobs lowered into ONE gate
    graph on its own GPU; synthetic ciphertexts (timing is value-independent; correctness of the same graph is
    tests/test_gpu_multiply.py).  Returns this rank's seconds per pool run; main() takes the MAX over ranks."""
    from spf_amd.gate_pool import circuit_jobs_as_one_graph, lpt_shards
    from spf_amd.mux_circuits import parse_mux_circuit
    circuit = parse_mux_circuit(open(os.path.join(DATA_DIR, "mux_multiplier_n8_m8.bincode"), "rb").read())
    n_jobs = per_gpu * world
    mine = lpt_shards([circuit.metrics()["mux_gates"]] * n_jobs, world)[rank]
    rng = np.random.default_rng(0x8008 + rank)
    cts = rng.integers(0, 1 << 64, size=(len(mine), 16, P.glwe_words), dtype=np.uint64)
    g, _ = circuit_jobs_as_one_graph(eng, circuit, cts)
    g.run()                       # plans, allocates, warms up
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        g.run()
    dt = (time.perf_counter() - t0) / reps
    st = g.stats()
    g.close()
    return {"_seconds": dt, "_units": per_gpu, "_rate_key": "multiplications_per_s",
            "_gates": per_gpu * (circuit.metrics()["mux_gates"] + 16),
            "multiplications": n_jobs, "per_gpu": per_gpu,
            "levels": st["levels"], "launches_per_rank": st["launches"],
            "block": "mux_circuits unsigned_multiplier(8,8): 3228 CMUX, depth 126, 16 circuit bootstraps"}


def _bench_mul32_pool(eng, P, rank, world, per_gpu=4):
    """BASELINE config 5: 32 x 32-bit encrypted multiplications via mux_circuits, one gate pool job = one
    multiplication built exactly as `append_uint_multiply` does (parasol_runtime/src/circuits/mul.rs:75-200): 64 input
    conversions, four `unsigned_multiplier(16, 16)` blocks (the reference's blob), 128 conversions of the partial
    products, `gradeschool_reduce(32, 32)`: ~127 k CMUX + 192 circuit bootstraps, 620+ levels.  `per_gpu` jobs per GPU
    (weak scaling), dealt by spf_amd.gate_pool, each rank's jobs in one graph on its GPU.  Synthetic ciphertexts;
    correctness of the same graph: tests/test_gpu_multiply.py::test_config5_encrypted_multiply_32x32."""
    from spf_amd import FheCircuit, ValueKind
    from spf_amd.gate_pool import lpt_shards
    from spf_amd.mux_circuits import GraphBuilder, append_uint_multiply, parse_mux_circuit
    blk16 = parse_mux_circuit(open(os.path.join(DATA_DIR, "mux_multiplier_n16_m16.bincode"), "rb").read())
    n_jobs = per_gpu * world
    mine = lpt_shards([1.0] * n_jobs, world)[rank]
    rng = np.random.default_rng(0x3232 + rank)
    t_build = time.perf_counter()
    g = FheCircuit(eng)
    builder = GraphBuilder(g)
    for _ in mine:
        sel = [builder.to_ggsw(g.add_input(ValueKind.GLWE1, rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64)))
               for _ in range(64)]
        for n in append_uint_multiply(builder, sel[:32], sel[32:], lambda a, b: {(16, 16): blk16}[(a, b)]):
            g.add_output(n, ValueKind.GLWE1)
    t_build = time.perf_counter() - t_build
    g.run()                       # plans, allocates (4 GB of GLWE per job), warms up
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        g.run()
    dt = (time.perf_counter() - t0) / reps
    st = g.stats()
    g.close()
    cmux = 4 * blk16.metrics()["mux_gates"] + 9104
    one = None
    if per_gpu > 1 and rank == 0:   # ONE multiplication alone on the GPU (the latency of config 5's circuit)
        g1 = FheCircuit(eng)
        b1 = GraphBuilder(g1)
        sel = [b1.to_ggsw(g1.add_input(ValueKind.GLWE1, rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64))) for _ in range(64)]
        for n in append_uint_multiply(b1, sel[:32], sel[32:], lambda a, b: {(16, 16): blk16}[(a, b)]):
            g1.add_output(n, ValueKind.GLWE1)
        g1.run()
        t0 = time.perf_counter()
        g1.run()
        one = {"ms_per_graph_run": round((time.perf_counter() - t0) * 1e3, 3), "launches": g1.stats()["launches"]}
        g1.close()
    return {"one_multiplication_per_graph": one,
            "_seconds": dt, "_units": per_gpu, "_rate_key": "multiplications_per_s", "_gates": per_gpu * (cmux + 192),
            "multiplications": n_jobs, "per_gpu": per_gpu,
            "cmux_per_multiplication": cmux, "circuit_bootstraps_per_multiplication": 192,
            "nodes": st["nodes"], "levels": st["levels"], "launches_per_rank": st["launches"],
            "graph_build_s": round(t_build, 2)}


def _bench_add32(eng, P, K, dev, gen, DevArray, torch, keys_loaded):
    """BASELINE config 3: K independent 32-bit additions through the reference's adder — `mux_circuits::add::
    ripple_carry_adder(32, 32, false)` (1 679 CMUX in 64 levels: one multiplexer per BDD node of every sum bit,
    add.rs:13-58 + lib.rs:358-445, rebuilt by spf_amd.mux_circuits) fed like `add_circuit` does (circuits/add.rs:10-32:
    64 x SampleExtract -> KeyswitchL1toL0 -> CircuitBootstrap) — as ONE `FheCircuit` run: one input copy, all levels
    enqueued back to back on one stream, one output copy.  `compact` is r01's hand-shared adder (192 CMUX + 32 Not in
    68 levels), kept for comparison.  Synthetic ciphertexts (timing is value-independent; correctness:
    tests/test_gpu_graph.py::test_encrypted_add_32_via_mux_circuits_adder / test_encrypted_add_32_as_one_graph)."""
    import spf_amd
    from spf_amd import FheOp, ValueKind
    from spf_amd.gate_pool import circuit_jobs_as_one_graph
    from spf_amd.mux_circuits import ripple_carry_adder
    rng = np.random.default_rng(0xADD32)

    def timed(g, reps=3):
        g.run()                       # plans, allocates, warms up
        t0 = time.perf_counter()
        for _ in range(reps):
            g.run()
        dt = (time.perf_counter() - t0) / reps
        st = g.stats()
        g.close()
        return dt, st

    adder = ripple_carry_adder(32, 32, False)
    cts = rng.integers(0, 1 << 64, size=(K, 64, P.glwe_words), dtype=np.uint64)
    g, _ = circuit_jobs_as_one_graph(eng, adder, cts)
    dt, st = timed(g)
    gates = adder.metrics()["mux_gates"] + 64
    # what the chip does with config 3 when it is USED: 4 and 16 additions per graph (64 conversions fill a quarter of the CUs)
    by_k = {str(K): {"ms_per_graph_run": round(dt * 1e3, 3), "adds_per_s": round(K / dt, 2)}}
    for k2 in (4, 16):
        if k2 == K:
            continue
        g2, _ = circuit_jobs_as_one_graph(eng, adder, rng.integers(0, 1 << 64, size=(k2, 64, P.glwe_words), dtype=np.uint64))
        d2, s2 = timed(g2, reps=2)
        by_k[str(k2)] = {"ms_per_graph_run": round(d2 * 1e3, 3), "adds_per_s": round(k2 / d2, 2), "launches": s2["launches"]}
    out = {"adds_per_graph": K, "circuit": f"mux_circuits ripple_carry_adder(32,32,false): {adder.metrics()['mux_gates']} CMUX, "
                                             f"depth {adder.depth()}, 64 circuit bootstraps",
           "ms_per_graph_run": round(dt * 1e3, 3), "adds_per_s": round(K / dt, 2), "gates_per_s": round(K * gates / dt, 1),
           "nodes": st["nodes"], "levels": st["levels"], "launches": st["launches"],
           "by_adds_per_graph": by_k,
           "note": "wall time of spf_graph_run: H2D of 64 GLWE inputs per add, all levels, D2H of 33 GLWE outputs per add"}

    g = spf_amd.FheCircuit(eng)
    for _ in range(K):
        sel = []
        for _ in range(64):
            x = g.add_input(ValueKind.GLWE1, rng.integers(0, 1 << 64, size=P.glwe_words, dtype=np.uint64))
            x = g.add_op(FheOp.SampleExtract, [x], 0)
            x = g.add_op(FheOp.KeyswitchL1toL0, [x])
            sel.append(g.add_op(FheOp.CircuitBootstrap, [x]))
        ga, gb = sel[:32], sel[32:]
        zero = g.add_trivial(ValueKind.GLWE1, 0)
        one = g.add_trivial(ValueKind.GLWE1, 1)
        carry = zero
        for i in range(32):
            ncarry = g.add_op(FheOp.Not, [carry])
            l1 = [g.add_op(FheOp.CMux, [gb[i], lo, hi]) for lo, hi in
                  [(carry, ncarry), (ncarry, carry), (zero, carry), (carry, one)]]
            g.add_output(g.add_op(FheOp.CMux, [ga[i], l1[0], l1[1]]), ValueKind.GLWE1)
            carry = g.add_op(FheOp.CMux, [ga[i], l1[2], l1[3]])
        g.add_output(carry, ValueKind.GLWE1)
    dt2, st2 = timed(g)
    out["compact"] = {"circuit": "192 CMUX + 32 Not, carries shared between sum and carry-out", "ms_per_graph_run": round(dt2 * 1e3, 3),
                      "levels": st2["levels"], "launches": st2["launches"]}
    return out


if __name__ == "__main__":
    sys.exit(main())
