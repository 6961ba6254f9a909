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
