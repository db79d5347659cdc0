"""Utterance sharding across the GPUs of one node (one process per GPU).

Each OfflineStream is an independent utterance (OfflineRecognizer.cs:192-197) and nothing
on the path crosses streams except the shared padding length and the first-emission
context switch, both of which the reference defines per GetResults batch.  So a shard is
decoded exactly as the reference would decode it as its own batch; there is NO data-path
collective.  torch.distributed (RCCL on GPU, gloo in the CPU tests) is used only for the
timing barrier, the max-over-ranks and the KB-sized gather of token lists.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def batches_of(lo: int, hi: int, batch: int) -> List[Tuple[int, int]]:
    """The GetResults batches of one shard: consecutive runs of `batch` utterances, [(first id, count)].  A shard is
    decoded exactly as the reference would decode these lists (padding length and the first-emission context switch are
    per batch, OfflineRecognizer.cs:213-216,278-286), so the same (total, world, batch) always gives the same batches
    whichever process runs them."""
    if batch <= 0:
        raise ValueError("batch must be positive")
    return [(a, min(batch, hi - a)) for a in range(lo, hi, batch)]


def gather_results(dist, local: Sequence, world: int, rank: int) -> List:
    """Concatenate per-rank result lists in rank order on every rank (host-side, tiny)."""
    if dist is None or world == 1:
        return list(local)
    bucket = [None] * world
    dist.all_gather_object(bucket, list(local))
    out = []
    for part in bucket:
        out.extend(part)
    return out


def max_over_ranks(dist, value: float, device=None) -> float:
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
