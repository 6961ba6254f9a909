"""bench.py prints ONE JSON line with the fields the driver's contract names, and its parity sample
(the CPU-baseline leg) agrees with the GPU word for word."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "512",
                        "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)]:
        assert isinstance(d[key], typ), (key, d[key])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "PBS/s" and d["dtype"] == "f64" and "workload" in d["config"]
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    # "fp64": the f64 work runs on the VALU, whose peak equals the MFMA-f64 peak (VERDICT r1 asked for the relabel)
    assert roof["bound"] in ("hbm", "mfma", "fp64") and 0.0 < roof["frac"] < 1.0
    # the kernel named is the one that ran: 512 ciphertexts (two per CU) take the two-ciphertext paired shape, not the
    # four-ciphertext throughput kernel
    assert roof["kernel"].startswith("blind_rotate2p2_kernel"), roof["kernel"]
    for leg in ("gate", "cmux", "circuit_bootstrap", "add32", "pcie_inclusive", "device_group", "evaluation_pool"):
        assert isinstance(d.get(leg), dict), (leg, d.get("leg_errors"))
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:500]     # nothing else on stdout (librccl's banner goes to stderr)
    grp = d["device_group"]
    assert grp["members"] == 2 and grp["same_words_as_single_context"] is True and grp["key_bytes_replicated_per_member"] > 0
    assert d["pcie_inclusive"]["same_words_as_device_path"] is True
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # traffic / busy fractions are this run's own rocprofv3 --pmc passes when the profiler is on the box, else the
    # committed passes (attached only for the same kernel and batch — not this batch size — so then traffic is null)
    if roof.get("counters"):
        assert roof["traffic"] == roof["counters"]["hbm_bytes_per_launch"] > 0
        assert 0.0 < roof["counters"]["valu_busy_frac"] < 1.0
    else:
        assert "live_counters" in (d.get("leg_errors") or {}) or roof["traffic"] is None
    cpu = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "cpus_visible", "cfs_quota_cpus"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0
    assert cpu["cores"] <= cpu["cpus_visible"] and (cpu["cfs_quota_cpus"] is None or cpu["cores"] <= cpu["cfs_quota_cpus"] + 1)
    assert cpu["gpu_outputs_bit_equal_on_sample"] is True


@pytest.mark.gpu
def test_single_process_bench_line_over_a_device_group():
    """`bench.py --gpus N --single-process`: the same line shape from ONE host process driving a device group (here two members
    on the one GPU), keys replicated inside the library."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0,0",
                        "--steps", "1", "--warmup", "1", "--batch", "256", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1, r.stdout[:500]
    d = json.loads(out[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "PBS/s" and d["value"] > 0
    assert d["config"]["devices"] == [0, 0] and d["config"]["global_batch"] == 512
    assert d["rccl"]["members"] == 2 and d["rccl"]["broadcast_bytes"] > 0 and d["rccl"]["world_size"] >= 1
    assert d["pcie_inclusive"]["same_words_as_device_path"] is True
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
