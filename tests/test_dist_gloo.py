"""world_size-2 CPU tests (gloo) of the anchor-pair sharding + score gather used on N GPUs.

The compute inside each rank is a stand-in (the CPU oracle, test-only) because there is no GPU here; what is under test is
loco_hd_amd.dist: the anchor-binned partition rule (every rank computes the same one without communication), the
permutation that travels with the scores, the gather, and the restore to anchor-pair order (output i <-> anchor pair i).
The same rule inside the library's kernels and a real DeviceSession under torch.distributed are in tests/test_gpu_dist.py."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
N_ATOMS = 120


def _workload():
    rng = np.random.default_rng(7)
    n = N_ATOMS
    xa, xb = rng.uniform(-12, 12, (n, 3)), rng.uniform(-12, 12, (n, 3))
    ca, cb = rng.integers(0, 4, n).astype(np.int32), rng.integers(0, 4, n).astype(np.int32)
    pairs = np.stack([rng.integers(0, n, 257), rng.integers(0, n, 257)], 1).astype(np.int64)  # odd count: uneven shards
    return xa, xb, ca, cb, pairs


def _score_fn():
    sys.path.insert(0, str(ROOT))
    from oracle import oracle as orc

    xa, xb, ca, cb, _ = _workload()
    lchd = orc.LoCoHD(["a", "b", "c", "d"], orc.WeightFunction("uniform", [3.0, 10.0]), n_of_threads=1)
    tag = np.zeros(len(xa), dtype=np.int32)

    def fn(anchor_slice):
        out = lchd.from_arrays(xa, ca, tag, xb, cb, tag, anchor_slice.numpy(), 10.0)
        return torch.tensor(out, dtype=torch.float64)

    return fn


def _worker(rank, world, port, result_file, partition):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    from loco_hd_amd.dist import score_sharded

    anchors = torch.from_numpy(_workload()[4])
    seen = []
    fn = _score_fn()

    def counting(sub):
        seen.append(sub.clone())
        return fn(sub)

    full = score_sharded(counting, anchors, world, rank, n_atoms_a=N_ATOMS, partition=partition)
    if partition == "anchor":  # the two ranks' side-A anchors do not interleave: rank 0 has the low atoms, rank 1 the high ones
        mine = torch.cat(seen)[:, 0]
        edge = torch.tensor([int(mine.max()) if rank == 0 else int(mine.min())])
        edges = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(edges, edge)
        assert int(edges[0]) <= int(edges[1])
    if rank == 0:
        np.save(result_file, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from loco_hd_amd.dist import shard_bounds

    for n, world in ((257, 2), (10, 8), (0, 4), (1_000_000, 8), (7, 7)):
        covered = []
        for r in range(world):
            lo, hi, chunk = shard_bounds(n, world, r)
            assert 0 <= lo <= hi <= n and hi - lo <= chunk
            covered += list(range(lo, hi))
        assert covered == list(range(n))


def test_anchor_partition_rule():
    """Every pair belongs to exactly one rank, ranks are balanced for spread-out anchors, a rank's side-A anchors form one
    contiguous range of atoms, and degenerate lists (one anchor for every pair, out-of-range indices) stay well defined."""
    from loco_hd_amd.dist import select_shard, shard_rule

    rng = np.random.default_rng(3)
    for n_atoms, p, world in ((200_000, 300_000, 8), (1000, 5000, 3), (50, 7, 4), (10, 100, 1), (3000, 0, 2)):
        a = torch.from_numpy(np.stack([rng.integers(0, n_atoms, p), rng.integers(0, n_atoms, p)], 1).astype(np.int64)).reshape(-1, 2)
        rank_of_pair, counts = shard_rule(a, n_atoms, world)
        assert sum(counts) == p and len(counts) == world
        assert np.array_equal(np.bincount(rank_of_pair.numpy(), minlength=world), np.asarray(counts))
        if p >= 1000:
            assert max(counts) <= 1.05 * p / world + n_atoms / 1024 + 8, counts
        covered = []
        hi_prev = -1
        for r in range(world):
            sel, idx, c2 = select_shard(a, n_atoms, world, r)
            assert c2 == counts and len(idx) == counts[r] and torch.equal(sel, a[idx])
            covered.append(idx.numpy())
            if len(sel) and p >= 1000:  # (side-A bins: every rank owns a contiguous range of side-A anchors)
                assert int(sel[:, 0].min()) > hi_prev or int(sel[:, 0].min()) // max(n_atoms // 1024, 1) >= hi_prev // max(n_atoms // 1024, 1)
                hi_prev = int(sel[:, 0].max())
        assert np.array_equal(np.sort(np.concatenate(covered)) if covered else np.zeros(0), np.arange(p))
    # KRas-scan shape (python_codes/kras_scan.py:46-52): one reference anchor against thousands.  Side A's histogram is
    # degenerate (one bin holds everything), so the rule bins by side B -- or, without the size of structure B, cuts the list
    # into contiguous slices; the reference's par_iter balances this list trivially (src/locohd.rs:545-557), and so must this.
    a = torch.from_numpy(np.stack([np.zeros(4000, np.int64), np.arange(4000)], 1))
    for n_b in (4000, None):
        rank_of_pair, counts = shard_rule(a, 5000, 4, n_b)
        assert sum(counts) == 4000 and max(counts) <= 1000 + 8, counts
        for r in range(4):
            sel, idx, c2 = select_shard(a, 5000, 4, r, n_atoms_b=n_b)
            assert c2 == counts and torch.equal(sel, a[idx]) and torch.equal(idx, (rank_of_pair == r).nonzero().reshape(-1))
    # both sides degenerate: one pair repeated -> contiguous slices
    a = torch.from_numpy(np.stack([np.full(999, 5), np.full(999, 7)], 1))
    _, counts = shard_rule(a, 10, 4, 10)
    assert sum(counts) == 999 and max(counts) - min(counts) <= 1, counts
    # indices outside the structure are clamped into the first / last bin (the scoring pass reports them)
    a = torch.tensor([[-5, 0], [10**9, 1], [3, 2]], dtype=torch.int64)
    _, counts = shard_rule(a, 100, 2)
    assert sum(counts) == 3


@pytest.mark.timeout(180)
@pytest.mark.parametrize("partition", ["anchor", "contiguous"])
def test_sharded_scores_match_single_process(tmp_path, partition):
    port = 29500 + (os.getpid() % 2000) + (7 if partition == "anchor" else 0)
    out = tmp_path / "scores.npy"
    mp.spawn(_worker, args=(2, port, str(out), partition), nprocs=2, join=True)
    got = np.load(out)
    want = _score_fn()(torch.from_numpy(_workload()[4])).numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


def test_partition_cache_follows_the_tensor():
    """select_shard(cache=True): the partition of an unchanged pair-list tensor is computed once; writing to the tensor (torch's
    version counter) or passing another tensor makes it be computed again."""
    import torch

    from loco_hd_amd import dist as D

    D.clear_shard_cache()
    g = torch.Generator().manual_seed(7)
    anchors = torch.randint(0, 5000, (20000, 2), generator=g, dtype=torch.int64)
    first = D.select_shard(anchors, 5000, 4, 1, n_atoms_b=5000, cache=True)
    again = D.select_shard(anchors, 5000, 4, 1, n_atoms_b=5000, cache=True)
    assert again[0] is first[0] and again[1] is first[1]
    plain = D.select_shard(anchors, 5000, 4, 1, n_atoms_b=5000)
    assert torch.equal(plain[0], first[0]) and torch.equal(plain[1], first[1]) and plain[2] == first[2]
    other_rank = D.select_shard(anchors, 5000, 4, 2, n_atoms_b=5000, cache=True)
    assert other_rank[0] is not first[0] and other_rank[2] == first[2]
    anchors[:10000, 0] = 0  # in place: the cached partition no longer describes the list
    changed = D.select_shard(anchors, 5000, 4, 1, n_atoms_b=5000, cache=True)
    assert changed[0] is not first[0]
    want = D.select_shard(anchors, 5000, 4, 1, n_atoms_b=5000)
    assert torch.equal(changed[0], want[0]) and torch.equal(changed[1], want[1])
    clone = anchors.clone()
    assert D.select_shard(clone, 5000, 4, 1, n_atoms_b=5000, cache=True)[0] is not changed[0]
    # least-recently-used eviction with room for max(16, 2 x world) entries: 40 emulated ranks all stay cached ...
    D.clear_shard_cache()
    small = torch.randint(0, 500, (4000, 2), generator=g, dtype=torch.int64)
    firsts = [D.select_shard(small, 500, 40, r, n_atoms_b=500, cache=True) for r in range(40)]
    assert all(D.select_shard(small, 500, 40, r, n_atoms_b=500, cache=True)[0] is firsts[r][0] for r in range(40))
    # ... a recently used entry survives newcomers that evict older ones, and a session's entries go with clear_shard_cache(session)
    D.clear_shard_cache()
    keep = D.select_shard(small, 500, 4, 0, n_atoms_b=500, cache=True)
    for k in range(20):
        D.select_shard(small.clone(), 500, 4, 0, n_atoms_b=500, cache=True)
        assert D.select_shard(small, 500, 4, 0, n_atoms_b=500, cache=True)[0] is keep[0]
    assert len(D._PLAN_CACHE) <= 16
    D.clear_shard_cache(session=object())  # (no entry belongs to it)
    assert D.select_shard(small, 500, 4, 0, n_atoms_b=500, cache=True)[0] is keep[0]
    D.clear_shard_cache(session=None)
    assert D.select_shard(small, 500, 4, 0, n_atoms_b=500, cache=True)[0] is not keep[0]
