"""The multi-GPU code path on the ONE GPU of the test box (task 1b/1c).

  * RCCL, world_size 1: `torch.distributed` with backend nccl (== RCCL on ROCm) initialised exactly as bench.py
    does, barrier with device_ids, `dist.broadcast` INTO the engine's own key memory (the four `spf_key_blob`
    views of spf_amd.sharding.key_blob_tensors), commit, then bootstrap and circuit-bootstrap against the oracle.
    What replaces the reference's shape here is the shared `Arc<ComputeKey>` of one process
    (parasol_runtime/src/circuit_processor/mod.rs:201-209, crypto/keys.rs:306-318): one key replica per GPU.
  * `python3 bench.py --gpus 2 --backend gloo` from a bare shell (no launcher, WORLD_SIZE unset): bench.py
    starts its own two ranks as child processes; both share the one GPU.

Each case runs in a child process: a process group belongs to a process, and the test runner keeps its own GPU
context for the other tests.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

_WS1 = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SPF_ROOT"])
import torch
import torch.distributed as dist
import oracle as O
import spf_amd
from spf_amd.sharding import key_blob_tensors, replicate_keys, max_over_ranks, shard_range
from tests.util import to_engine_params

n = 12
P = O.DEFAULT_128.replace(lwe_n=n)
ks = O.gen_keyset(0x5EED0001, P)
r = O.Rng(0x7A11)
ak, ssk = O.gen_auto_key_fft(r, ks.glwe_sk, P), O.gen_ssk_fft(r, ks.glwe_sk, P)

os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ.setdefault("MASTER_PORT", os.environ["SPF_PORT"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.barrier(device_ids=[0])
eng = spf_amd.Engine(to_engine_params(P), device=0)
blobs = key_blob_tensors(eng, dev)
host = [ks.bsk_fft.view(np.uint8), ks.ksk.view(np.uint8), ak.view(np.uint8), ssk.view(np.uint8)]
for b, h in zip(blobs, host):
    assert b.numel() == h.size, (b.numel(), h.size)
    b.copy_(torch.from_numpy(np.ascontiguousarray(h).reshape(-1)))
secs, nbytes = replicate_keys(eng, blobs, dist, src=0)       # dist.broadcast x 4 over RCCL, then commit
assert nbytes == sum(h.size for h in host)
for b, h in zip(blobs, host):                                  # the broadcast left the key bytes in place
    assert torch.equal(b.cpu(), torch.from_numpy(np.ascontiguousarray(h).reshape(-1)))
assert max_over_ranks(3.5, dist, device=dev) == 3.5            # the bench's timing reduction, on RCCL
b0, e0 = shard_range(5, 0, 1)
lwe = np.random.default_rng(5).integers(0, 1 << 64, (e0 - b0, n + 1), dtype=np.uint64)
got = eng.circuit_bootstrap_pbs(lwe)
for i in range(len(lwe)):
    assert np.array_equal(got[i], O.cbs_pbs(lwe[i], ks.bsk_fft, P)), i
lwe1 = np.random.default_rng(6).integers(0, 1 << 64, (3, P.N + 1), dtype=np.uint64)
l0 = eng.keyswitch_lwe_l1_lwe_l0(lwe1)                         # the committed keyswitch key (byte planes derived locally)
for i in range(3):
    assert np.array_equal(l0[i], O.keyswitch_lwe(lwe1[i], ks.ksk, P.N, n, P.ks_radix_log, P.ks_count)), i
gg = eng.circuit_bootstrap(lwe[:2])                            # automorphism + scheme-switch keys from the broadcast
for i in range(2):
    exp = O.circuit_bootstrap(lwe[i], ks.bsk_fft, ak, ssk, P)
    assert np.array_equal(gg[i].view(np.float64).reshape(-1), exp.view(np.float64).reshape(-1)), i
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("RCCL_WS1_OK", round(secs, 4), nbytes)
'''


def _bare_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env["SPF_ROOT"] = ROOT
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_rccl_world_size_1_key_broadcast_then_bootstrap():
    env = _bare_env()
    env["SPF_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, "-c", _WS1], capture_output=True, text=True, timeout=540, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RCCL_WS1_OK" in r.stdout, r.stdout[-1500:]


@pytest.mark.timeout(900)
def test_bench_self_launches_two_gloo_ranks_on_one_gpu():
    """`python3 bench.py --gpus 2 --backend gloo` with no launcher: rc 0, one JSON line from rank 0, two ranks seen by
    the process group, every leg of the line present on two ranks (the job-sharded gate pools included)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "1", "--batch", "512"], capture_output=True, text=True, timeout=840, cwd=ROOT,
                       env=_bare_env())
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 1024 and d["scaling"] == "weak"
    assert d["rccl"]["backend"] == "gloo" and d["rccl"]["world_size"] == 2 and d["rccl"]["broadcast_GBs"] > 0
    assert d["cpu_baseline"] is None                      # rank 0 at N = 1 only
    assert "leg_errors" not in d, d.get("leg_errors")
    for leg in ("gate", "cmux", "circuit_bootstrap", "mul8_gate_pool", "mul32_gate_pool"):
        assert isinstance(d.get(leg), dict), leg
    assert d["mul8_gate_pool"]["multiplications"] == 16 and d["mul32_gate_pool"]["multiplications"] == 8
