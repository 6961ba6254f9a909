"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Bootstraps are independent units, so the path shards with no data-path collective: rank r owns
a contiguous range of the batch and holds a full replica of the evaluation keys.  The only
collective is the one-time broadcast of the keys from the rank that holds them; timing is
reduced with MAX over ranks.  Everything here works on CPU tensors with the gloo backend too
(that is how it is tested without GPUs).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `total` units for `rank`: ceil(total/world) per rank, the tail
    ranks may get fewer (or none)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-total // world)
    begin = min(total, rank * per)
    return begin, min(total, begin + per)


def shard_sizes(total: int, world: int) -> List[int]:
    return [e - b for b, e in (shard_range(total, r, world) for r in range(world))]


def broadcast_keys(blobs: Sequence, dist, src: int = 0) -> None:
    """In-place broadcast of the key blobs (uint8 tensors viewing each rank's key memory)."""
    for b in blobs:
        dist.broadcast(b, src=src)


class _DevArray:
    """zero-copy view of a raw device pointer (CUDA array interface), for torch.as_tensor"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False),
                                         "version": 3, "strides": None}


def key_blob_tensors(engine, device) -> List:
    """The four ComputeKey fields of `engine` (bootstrap, keyswitch, automorphism, scheme switch:
    parasol_runtime/src/crypto/keys.rs:306-318) as uint8 torch tensors that ALIAS the engine's own HBM
    (spf_key_blob): what the RCCL broadcast writes into.  Call `replicate_keys` (or spf_key_blob_commit per
    blob) after filling them."""
    import torch
    out = []
    for which in (0, 1, 2, 3):
        ptr, nbytes = engine.key_blob(which)
        out.append(torch.as_tensor(_DevArray(ptr, nbytes), device=device))
    return out


def replicate_keys(engine, blobs: Sequence, dist=None, src: int = 0) -> Tuple[float, int]:
    """One-time key replication: broadcast the four blobs from rank `src` (RCCL over xGMI when the process
    group's backend is nccl; skipped without a process group), then commit them on this rank's GPU (derives the
    keyswitch byte planes, drops cached constants).  Returns (seconds spent in the broadcast, bytes per rank)."""
    import time

    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if dist is not None and dist.is_initialized():
        broadcast_keys(blobs, dist, src=src)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for which in range(len(blobs)):
        engine.key_blob_commit(which)
    return dt, sum(int(b.numel()) for b in blobs)


def max_over_ranks(value: float, dist, device=None) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_shards(local, total: int, dist, rank: int, world: int):
    """Reassemble per-rank outputs (first dimension = units) in rank order on every rank."""
    import torch
    per = -(-total // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:n] for p, n in zip(parts, shard_sizes(total, world))], dim=0)
