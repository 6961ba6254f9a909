"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Exercises the
same helpers bench.py uses (spf_amd.sharding): contiguous batch sharding with no data-path
collective, one-time key broadcast, MAX-over-ranks timing, ordered reassembly.  The per-shard
compute stands in with the CPU oracle (test infrastructure) because the product has no CPU
path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spf_amd.sharding import broadcast_keys, gather_shards, max_over_ranks, shard_range, shard_sizes


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
                assert e0 == b1 and b0 <= e0
            assert sum(shard_sizes(total, world)) == total
    assert shard_range(65536, 3, 8) == (3 * 8192, 4 * 8192)   # BASELINE config 4
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = O.DEFAULT_128.replace(lwe_n=3)
        # rank 0 owns the keys; everyone else starts with zeros and receives the broadcast
        if rank == 0:
            keys = O.gen_keyset(0xBEEF, P)
            bsk = torch.from_numpy(keys.bsk_fft.view(np.uint8).copy())
            ksk = torch.from_numpy(keys.ksk.view(np.uint8).copy())
        else:
            bsk = torch.zeros(P.lwe_n * P.ggsw_fft_len * 16, dtype=torch.uint8)
            ksk = torch.zeros(P.N * P.ks_count * (P.lwe_n + 1) * 8, dtype=torch.uint8)
        broadcast_keys([bsk, ksk], dist, src=0)
        bsk_np = bsk.numpy().view(np.complex128)
        # every rank sees the same global batch definition and takes its contiguous shard
        lwe = np.random.default_rng(7).integers(0, 1 << 64, (total, P.lwe_n + 1), dtype=np.uint64)
        b, e = shard_range(total, rank, world)
        local = np.stack([O.cbs_pbs(lwe[i], bsk_np, P) for i in range(b, e)]) if e > b else \
            np.zeros((0, P.glwe_len), dtype=np.uint64)
        full = gather_shards(torch.from_numpy(local.view(np.int64)), total, dist, rank, world)
        t = max_over_ranks(1.0 + rank, dist)
        if rank == 0:
            exp = np.stack([O.cbs_pbs(x, bsk_np, P) for x in lwe])
            q.put((bool(np.array_equal(full.numpy().view(np.uint64), exp)), t, int(bsk.sum())))
        else:
            q.put((None, t, int(bsk.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_shard_broadcast_gather():
    world, total = 2, 5          # ragged: 3 + 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert any(r[0] is True for r in res)            # rank 0: gathered == unsharded run
    assert all(abs(r[1] - 2.0) < 1e-12 for r in res)  # MAX over ranks of (1, 2)
    assert res[0][2] == res[1][2] != 0               # both ranks hold the same key bytes


def _bare_env():
    return {k: v for k, v in os.environ.items()
            if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}


def test_bench_rejects_a_launcher_mismatch():
    """--gpus that disagrees with the launcher's WORLD_SIZE is an error (rc 2), before any GPU work."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _bare_env()
    env.update({"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, cwd=root, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


@pytest.mark.timeout(300)
def test_bench_self_launch_starts_its_ranks_as_children():
    """`python3 bench.py --gpus 2` with no launcher spawns torch.distributed.run with two ranks and returns their
    status.  Without a GPU (this test runs on CPU) each rank must stop with "no GPU visible" — there is no CPU path —
    and the parent must report the failure instead of hanging or printing a line."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check of the launcher plumbing; the GPU form is tests/test_gpu_rccl.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1"],
                       capture_output=True, text=True, timeout=280, cwd=root, env=_bare_env())
    assert r.returncode != 0
    assert r.stderr.count("no GPU visible") >= 2, r.stderr[-2000:]       # both child ranks got as far as the GPU check
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
