"""CPU checks of bench.py's workload generators, its strong-scaling partition of C3 and its roofline arithmetic (the parts the
driver's scaling runs rely on but no GPU test looks at)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_c3_tiles_cover_every_decoy_pair_once(world):
    nd = 50
    shares = bench.tile_structure_pairs(nd, world)
    assert len(shares) == world
    flat = [p for s in shares for p in s]
    assert len(flat) == nd * (nd - 1) // 2 and len(set(flat)) == len(flat)
    assert all(0 <= a < b < nd for a, b in flat)
    sizes = [len(s) for s in shares]
    assert max(sizes) <= 1.45 * (sum(sizes) / world) + 1, sizes  # (whole tiles are dealt: balance within a tile's worth)
    if world == 8:  # the point of tiling: a rank touches half of the decoys (two of four groups), shares within a group's rounding
        touched = [len({d for p in s for d in p}) for s in shares]
        assert max(touched) <= 26, touched
        assert max(sizes) <= 1.12 * (sum(sizes) / world), sizes


def test_workload_generators_are_deterministic_and_shaped():
    w = bench.make_workload("c2a", 0, 30_000)
    assert w["pairs"].shape == (30_000, 2) and w["pairs"].dtype == np.int64 and w["n"] == 10_000 and w["C"] == 10
    assert np.array_equal(w["pairs"][:10_000, 0], np.arange(10_000)) and np.array_equal(w["pairs"][:10_000, 1], np.arange(10_000))
    assert sorted(w["pairs"][10_000:20_000, 1].tolist()) == list(range(10_000))  # later rounds: permutations of all atoms
    again = bench.make_workload("c2a", 0, 30_000)
    assert np.array_equal(w["pairs"], again["pairs"]) and np.array_equal(w["xyz_a"], again["xyz_a"])
    other = bench.make_workload("c2a", 1, 30_000)  # weak scaling: another rank scores other pairs of the same clouds
    assert np.array_equal(w["xyz_a"], other["xyz_a"]) and not np.array_equal(w["pairs"], other["pairs"])
    same = bench.make_workload("c2a", 1, 30_000, same_on_all_ranks=True)  # strong scaling: one list for the whole job
    assert np.array_equal(w["pairs"], same["pairs"])
    w5 = bench.make_workload("c5", 0, 5_000)
    assert w5["n"] == 200_000 and w5["C"] == 25 and w5["pairs"].shape == (5_000, 2) and w5["pairs"].max() < 200_000
    wd = bench.make_workload("c2b", 0, 0, n_atoms=1234)
    assert wd["n"] == 1234 and np.array_equal(wd["pairs"][:, 0], wd["pairs"][:, 1]) and len(wd["pairs"]) == 1234
    c3 = bench.make_c3(0, True)
    assert c3["nd"] == 50 and len(c3["spairs"]) == 1225 and all(len(d[0]) == 3000 for d in c3["decoys"])
    assert all((d[1][::3] == 0).all() and (d[1][1::3] > 0).all() for d in c3["decoys"][:3])  # every third point a "Cent" anchor


def test_roofline_block_arithmetic(tmp_path, monkeypatch):
    blk = bench.roofline_block("no_such_workload", "k_sweep", 9.6e9, 1.6)
    assert blk["bound"] == "valu" and blk["unit"] == "GB/s" and blk["peak"] == bench.HBM_PEAK_GBS
    assert blk["achieved"] == pytest.approx(9.6e9 / 1.6e-3 / 1e9) and blk["frac"] == pytest.approx(blk["achieved"] / blk["peak"])
    assert blk["traffic"] is None and "fabric_measured_frac" not in blk and "frac_step" not in blk
    blk = bench.roofline_block("no_such_workload", "k_sweep", 9.6e9, 1.6, launches_per_step=2, step_ms=4.0)
    assert blk["frac_step"] == pytest.approx(2 * 9.6e9 / 4.0e-3 / 8e12)
    # with a committed profile of the workload: measured traffic and the issue utilisation travel with the line
    prof = {"kernel": "k_sweep<12, 0, 0, true, false, false, true>", "traffic_bytes_per_launch": 2.7e9, "valu_issue_frac": 0.9, "binding": "valu issue",
            "avg_launch_ns_kernel_trace": 1.5e6}
    monkeypatch.setattr(bench, "profile_numbers", lambda w: prof)
    blk = bench.roofline_block("c2a", "k_sweep (its launch and the k_pair_meta record pass in front of it)", 9.6e9, 1.6)
    assert blk["traffic"] == 2.7e9 and blk["fabric_measured_frac"] == pytest.approx(2.7e9 / 1.5e-3 / 8e12)  # the PROFILED time, not the live 1.6 ms
    assert blk["valu_issue_frac"] == 0.9 and blk["binding"] == "valu issue"
    # a profile of another kernel family is not attached
    blk = bench.roofline_block("c2b", "k_env_rows2 (both structures' rows in one launch)", 5.6e9, 2.3)
    assert blk["traffic"] is None


def _run_bench(extra_env, *argv, timeout=180):
    import os
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python3 bench.py --gpus N` -- the form the driver uses for N = 1 -- must not die for N > 1 (round 5: rc 1, "launch with
    torch.distributed.run"): it starts N ranks as a child process.  Here the ranks only meet (gloo) and report."""
    import json

    p = _run_bench({"LCHD_BENCH_RENDEZVOUS_ONLY": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert p.returncode == 0, p.stderr[-2000:]
    assert "rank 0 of 2 up" in p.stderr and "rank 1 of 2 up" in p.stderr, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])  # rank 0's line is the last line of stdout, relayed by the parent
    assert line == {"rendezvous": "ok", "world": 2, "rank_sum": 1.0, "gpus": 2}


def test_bench_gpus_2_on_a_machine_with_fewer_gpus_fails_loudly_and_quickly():
    """Exactly the driver's command.  Without two GPUs (this container: none; a one-GPU box: one) every rank comes up, the rank
    without a device says so, the launcher takes the others down, and the exit code is not 0 -- no hang."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the command would run the real benchmark")
    p = _run_bench({}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert p.returncode != 0
    assert "rank 0 of 2 up" in p.stderr and "rank 1 of 2 up" in p.stderr, p.stderr[-2000:]
    assert "needs 2 visible devices" in p.stderr, p.stderr[-2000:]
    assert "{\"metric\"" not in p.stdout
