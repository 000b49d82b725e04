"""world_size-2 CPU test (gloo) of the anchor-pair sharding + score gather used on N GPUs.

The compute inside each rank is a stand-in (the CPU oracle, test-only) because there is no GPU here; what is
under test is loco_hd_amd.dist: slice bounds, padding, gather order (output i <-> anchor pair i)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _workload():
    rng = np.random.default_rng(7)
    n = 120
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


def _worker(rank, world, port, result_file):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    from loco_hd_amd.dist import score_sharded

    anchors = torch.from_numpy(_workload()[4])
    full = score_sharded(_score_fn(), anchors, world, rank)
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


@pytest.mark.timeout(180)
def test_sharded_scores_match_single_process(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    out = tmp_path / "scores.npy"
    mp.spawn(_worker, args=(2, port, str(out)), nprocs=2, join=True)
    got = np.load(out)
    want = _score_fn()(torch.from_numpy(_workload()[4])).numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
