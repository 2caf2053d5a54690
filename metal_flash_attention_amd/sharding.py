"""Batch x head sharding across GPUs -- the only multi-GPU structure this path has.

Every (batch, head) pair is an independent attention problem (the reference is single-headed and
turns multi-head into a stride change, AttentionKernelDescriptor.swift:37-41), so ranks own disjoint
contiguous ranges of the flattened batch x head axis and NO data-path collective exists.
torch.distributed is used only to line ranks up (barrier) and to take the max of the elapsed time.
"""
from __future__ import annotations

from typing import List, Tuple


def shard_range(total_units: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of the flattened batch x head units owned by `rank`: contiguous, balanced to
    within one unit, covering every unit exactly once."""
    if world_size <= 0 or not (0 <= rank < world_size) or total_units < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total_units, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def all_ranges(total_units: int, world_size: int) -> List[Tuple[int, int]]:
    return [shard_range(total_units, world_size, r) for r in range(world_size)]


def max_over_ranks(seconds: float, dist=None, device: str = "cpu") -> float:
    """Elapsed time of the slowest rank (the bench contract's timing rule)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
