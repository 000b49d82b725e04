"""Multi-GPU sharding of the anchor-pair list (one process per GPU, torch.distributed; backend "nccl" = RCCL).

Anchor pairs are independent (the reference's rayon par_iter body shares no mutable state,
/root/reference/src/locohd.rs:545-554), so the path has no exchange step: every rank holds both structures and the whole
pair list, scores its share, and the only communication is ONE gather of the score slices to rank 0 (8 B + 8 B of index per
pair: latency-bound on xGMI; no all-reduce).

Which share?  A rank that takes a contiguous slice of RANDOM pairs builds almost every environment of both structures
itself, so the environment phase does not shrink with the number of GPUs.  Binning the pairs by their side-A anchor
(`partition="anchor"`, the default) gives every rank ~1/world of side A's environments; the rule is a pure function of the
pair list, so all ranks agree on it without communication, and the original positions travel with the scores so that
output i still belongs to anchor pair i (order-preserving like the reference's indexed collect, src/locohd.rs:556).

    bin(p)  = floor(a_p * 1024 / n_atoms_a)                            a_p = side-A anchor of pair p (clamped into range)
    rank(b) = min(world - 1, floor(#pairs in bins < b * world / P))

A list whose side-A partition is unbalanced -- some rank would hold more than 1.25 P / world + 1 pairs: ONE reference anchor
against thousands, the call shape of /root/reference/python_codes/kras_scan.py:46-52 -- is binned by its side-B anchors instead,
and if that partition is unbalanced too (or `n_atoms_b` is not given), cut into contiguous slices rank(p) = floor(p * world / P):
the reference's par_iter balances any list (src/locohd.rs:545-557), and so does this rule.

On CUDA tensors with a `DeviceSession` the partition runs in the library's own kernels (lchd_shard_plan_dev /
lchd_shard_select_dev / lchd_unshard_scores_dev: two launches and one wait); the torch expressions below are the same rule
for CPU tensors (the gloo tests) and the cross-check of the kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Tuple

SHARD_BINS = 1024


def shard_bounds(n_pairs: int, world: int, rank: int) -> Tuple[int, int, int]:
    """Contiguous slice [lo, hi) of rank `rank` (partition="contiguous"); every rank's slice is padded to `chunk` for the gather."""
    chunk = (n_pairs + world - 1) // world
    lo = min(rank * chunk, n_pairs)
    return lo, min(lo + chunk, n_pairs), chunk


def shard_rule(anchors, n_atoms_a: int, world: int, n_atoms_b: Optional[int] = None):
    """The partition rule in torch (any device): returns (rank_of_pair int64 [P], counts list[world])."""
    import torch

    p = anchors.shape[0]

    def plan_side(col, n_atoms):
        a = anchors[:, col].clamp(0, max(int(n_atoms) - 1, 0))
        bins = (a * SHARD_BINS) // int(n_atoms)
        hist = torch.bincount(bins, minlength=SHARD_BINS)
        before = torch.cumsum(hist, 0) - hist
        rank_of_bin = torch.clamp((before * world) // max(p, 1), max=world - 1)
        counts = torch.zeros(world, dtype=torch.int64, device=anchors.device).index_add_(0, rank_of_bin, hist)
        balanced = int(counts.max()) * 4 * world <= 5 * p + 4 * world  # no rank holds more than 1.25 P / world + 1 pairs
        return rank_of_bin[bins], [int(v) for v in counts.tolist()], balanced

    rank_of_pair, counts, ok = plan_side(0, n_atoms_a)
    if not ok and n_atoms_b:
        rank_of_pair, counts, ok = plan_side(1, n_atoms_b)
    if not ok:  # contiguous slices of the pair list
        rank_of_pair = (torch.arange(p, dtype=torch.int64, device=anchors.device) * world) // max(p, 1)
        counts = [int(v) for v in torch.bincount(rank_of_pair, minlength=world).tolist()]
    return rank_of_pair, counts


# Partitions of pair lists that were seen before (select_shard(..., cache=True)).  The rule is a pure function of the list, so a
# caller that scores the SAME list again -- the frames of a trajectory, the rounds of a permutation test, the steps of bench.py --
# neither reads it twice per call nor launches the two partition kernels again.  An entry is valid for the very tensor object it
# was made from (kept alive by the entry) and only while that tensor has not been written to THROUGH TORCH (its version counter).
# Limits, by construction: a write that bypasses torch's versioning -- a kernel writing through data_ptr() (this library's own
# ctypes calls do), NumPy memory shared with a CPU tensor -- is NOT seen and the stale partition would put scores at the wrong
# positions: callers that fill pair lists that way pass cache=False (the default) or call clear_shard_cache().  Entries pin their
# tensors (the list, the selection, the positions) until they are evicted, cleared, or their session is closed
# (DeviceSession.close drops the entries of that session); eviction is least-recently-used with room for
# max(16, 2 x world) entries (one per emulated rank and session of bench.py --emulate-world).
_PLAN_CACHE: "dict" = {}
_PLAN_CACHE_MIN = 16


def clear_shard_cache(session=None) -> None:
    """Drop every remembered partition (session=None) or those made with one DeviceSession."""
    if session is None:
        _PLAN_CACHE.clear()
        return
    for key in [k for k in _PLAN_CACHE if k[5] == id(session)]:
        del _PLAN_CACHE[key]


def select_shard(anchors, n_atoms_a: int, world: int, rank: int, session=None, n_atoms_b: Optional[int] = None, cache: bool = False):
    """This rank's pairs: (sel_anchors [n][2], sel_index [n] = positions in the full list, counts of every rank).

    cache=True: the result is remembered per (tensor object, its version, world, rank, structure sizes, session) and returned
    again -- the same tensors, do not modify them -- while `anchors` is unchanged AS FAR AS TORCH CAN TELL (see the note above
    _PLAN_CACHE: writes through raw pointers are not detected)."""
    if cache:
        key = (id(anchors), world, rank, int(n_atoms_a), int(n_atoms_b or 0), id(session))
        hit = _PLAN_CACHE.get(key)
        if hit is not None and hit[0] is anchors and hit[1] == anchors._version:
            _PLAN_CACHE[key] = _PLAN_CACHE.pop(key)  # most recently used last
            return hit[2]
        res = select_shard(anchors, n_atoms_a, world, rank, session, n_atoms_b, cache=False)
        _PLAN_CACHE.pop(key, None)
        while len(_PLAN_CACHE) >= max(_PLAN_CACHE_MIN, 2 * world):
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))  # least recently used first
        _PLAN_CACHE[key] = (anchors, anchors._version, res)
        return res
    import torch

    p = anchors.shape[0]
    if session is not None and anchors.is_cuda and p > 0:
        from . import _native as N

        assert anchors.dtype == torch.int64 and anchors.is_contiguous()
        counts = (C.c_int64 * world)()
        nb = int(n_atoms_b) if n_atoms_b else 0
        N.check(N.lib().lchd_shard_plan_dev(session._ctx, C.c_void_p(anchors.data_ptr()), p, int(n_atoms_a), nb, world, counts))
        counts = [int(v) for v in counts]
        n = counts[rank]
        sel = torch.empty((n, 2), dtype=torch.int64, device=anchors.device)
        idx = torch.empty(n, dtype=torch.int64, device=anchors.device)
        N.check(N.lib().lchd_shard_select_dev(session._ctx, C.c_void_p(anchors.data_ptr()), p, int(n_atoms_a), nb, rank,
                                              C.c_void_p(sel.data_ptr()), C.c_void_p(idx.data_ptr())))
        return sel, idx, counts
    rank_of_pair, counts = shard_rule(anchors, n_atoms_a, world, n_atoms_b)
    idx = (rank_of_pair == rank).nonzero().reshape(-1)
    return anchors[idx].contiguous(), idx, counts


def gather_scores(local, gathered, world: int, rank: int, group=None, force_collective: bool = False, async_op: bool = False):
    """Gather equal-length vectors to rank 0: gathered[r*len : (r+1)*len] = rank r's `local`.
    With async_op=True the collective's work handle is returned (call .wait() before reusing `local`)."""
    import torch.distributed as dist

    if world == 1 and not force_collective:
        if gathered is not None:
            gathered[: local.numel()].copy_(local.reshape(-1))
        return None
    n = local.numel()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no gather of device tensors: staged through the host (rehearsals of the multi-rank path on one GPU or on CPU
        # ranks; RCCL -- backend "nccl" -- gathers device to device).  `.cpu()` waits for the scoring pass on the current stream.
        import torch

        host = local.reshape(-1).cpu()
        got = [torch.empty(n, dtype=local.dtype) for _ in range(world)] if rank == 0 else None
        dist.gather(host, gather_list=got, dst=0, group=group)
        if rank == 0:
            gathered[: world * n].copy_(torch.cat(got))
        return _DoneWork() if async_op else None
    chunks = [gathered[r * n:(r + 1) * n] for r in range(world)] if rank == 0 else None
    work = dist.gather(local.reshape(-1), gather_list=chunks, dst=0, group=group, async_op=async_op)
    return work if async_op else None


class _DoneWork:
    """work handle of a collective that had already completed when it was handed out"""

    def wait(self):
        return True


def unshard(gathered, counts: List[int], stride: int, n_pairs: int, out=None, session=None):
    """Rank 0: gathered is [world][2][stride] float64 (scores, then the pair positions as int64 bit patterns)."""
    import torch

    world = len(counts)
    if out is None:
        out = torch.empty(n_pairs, dtype=torch.float64, device=gathered.device)
    if session is not None and gathered.is_cuda and n_pairs > 0:
        from . import _native as N

        cnt = (C.c_int64 * world)(*counts)
        N.check(N.lib().lchd_unshard_scores_dev(session._ctx, C.c_void_p(gathered.data_ptr()), cnt, world, int(stride),
                                                C.c_void_p(out.data_ptr()), int(n_pairs)))
        return out
    g = gathered.reshape(world, 2, stride)
    for r, n in enumerate(counts):
        out[g[r, 1, :n].view(torch.int64)] = g[r, 0, :n]
    return out


def score_sharded(score_fn: Callable, anchors, world: int, rank: int, group=None, n_atoms_a: Optional[int] = None, session=None,
                  partition: str = "anchor", force_collective: bool = False, n_atoms_b: Optional[int] = None, cache_plan: bool = False):
    """Score this rank's share of `anchors` with `score_fn(anchor_subset [n][2]) -> 1-D float64 tensor [n]` and gather.

    `anchors` is the FULL [P][2] int64 tensor (same on every rank).  Returns the full [P] score tensor on rank 0 (output i
    belongs to anchor pair i) and None elsewhere.  partition="anchor" needs n_atoms_a (the size of structure A; with n_atoms_b
    a list that is degenerate on side A is binned by side B);
    "contiguous" is the plain slice (best when consecutive pairs share anchors already, e.g. (i, perm(i)) lists).
    With `session` (a DeviceSession on this rank's GPU) partition and restore run in the library's kernels.
    cache_plan=True: the partition of this very `anchors` tensor is computed once and reused while the tensor is unchanged."""
    import torch

    p = anchors.shape[0]
    dev = anchors.device
    if partition == "contiguous":
        lo, hi, chunk = shard_bounds(p, world, rank)
        local = torch.zeros(chunk, dtype=torch.float64, device=dev)
        if hi > lo:
            local[: hi - lo] = score_fn(anchors[lo:hi].contiguous())
        gathered = torch.empty(chunk * world, dtype=torch.float64, device=dev) if rank == 0 else None
        gather_scores(local, gathered, world, rank, group, force_collective)
        return gathered[:p] if rank == 0 else None
    if partition != "anchor":
        raise ValueError(f"unknown partition {partition!r}")
    if n_atoms_a is None:
        raise ValueError('partition="anchor" needs n_atoms_a')
    sel, idx, counts = select_shard(anchors, n_atoms_a, world, rank, session, n_atoms_b, cache=cache_plan)
    stride = max(max(counts), 1)
    local = torch.zeros((2, stride), dtype=torch.float64, device=dev)
    n = counts[rank]
    if n:
        local[0, :n] = score_fn(sel)
        local[1, :n] = idx.view(torch.float64)
    gathered = torch.empty(world * 2 * stride, dtype=torch.float64, device=dev) if rank == 0 else None
    gather_scores(local, gathered, world, rank, group, force_collective)
    if rank != 0:
        return None
    return unshard(gathered, counts, stride, p, session=session)
