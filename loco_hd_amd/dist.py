"""Multi-GPU sharding of the anchor-pair list (one process per GPU, torch.distributed; backend "nccl" = RCCL).

Anchor pairs are independent (the reference's rayon par_iter body shares no mutable state,
/root/reference/src/locohd.rs:545-554), so the pair list is cut into `world` contiguous slices, every rank
scores its slice against its own replica of the two structures, and the only communication is one gather of
the f64 score slices to rank 0 (<= 1 MB per rank at 10^6 pairs: latency-bound on xGMI; no all-reduce).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def shard_bounds(n_pairs: int, world: int, rank: int) -> Tuple[int, int, int]:
    """Contiguous slice [lo, hi) of rank `rank`; every rank's slice is padded to `chunk` for the gather."""
    chunk = (n_pairs + world - 1) // world
    lo = min(rank * chunk, n_pairs)
    return lo, min(lo + chunk, n_pairs), chunk


def gather_scores(local, gathered, world: int, rank: int, group=None, force_collective: bool = False, async_op: bool = False):
    """Gather equal-length score vectors to rank 0: gathered[r*len : (r+1)*len] = rank r's `local`.
    With async_op=True the collective's work handle is returned (call .wait() before reusing `local`)."""
    import torch.distributed as dist

    if world == 1 and not force_collective:
        if gathered is not None:
            gathered[: local.numel()].copy_(local)
        return None
    n = local.numel()
    chunks = [gathered[r * n:(r + 1) * n] for r in range(world)] if rank == 0 else None
    work = dist.gather(local, gather_list=chunks, dst=0, group=group, async_op=async_op)
    return work if async_op else None


def score_sharded(score_fn: Callable, anchors, world: int, rank: int, group=None):
    """Score anchors[lo:hi] on this rank with `score_fn(anchor_slice) -> 1-D float64 tensor` and gather.

    `anchors` is the FULL [P][2] int64 tensor (same on every rank).  Returns the full [P] score tensor on
    rank 0 (output i belongs to anchor pair i) and None elsewhere.
    """
    import torch

    p = anchors.shape[0]
    lo, hi, chunk = shard_bounds(p, world, rank)
    local = torch.zeros(chunk, dtype=torch.float64, device=anchors.device)
    if hi > lo:
        local[: hi - lo] = score_fn(anchors[lo:hi].contiguous())
    gathered = torch.empty(chunk * world, dtype=torch.float64, device=anchors.device) if rank == 0 else None
    gather_scores(local, gathered, world, rank, group)
    return gathered[:p] if rank == 0 else None
