"""Multi-GPU plumbing on ONE GPU: the library's partition kernels against the torch rule, the restore, `score_sharded` with a
real DeviceSession under torch.distributed (backend nccl = RCCL, world size 1, the collective forced), every rank of an
emulated world on the same device, and the single-process device group of the C ABI (two contexts on device 0)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT = 1e-11


@pytest.fixture(scope="module")
def setup(oracle):
    import torch

    import loco_hd_amd as lh
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(5)
    n, c = 4000, 9
    side = (n / 0.04) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, c, n).astype(np.int32), rng.integers(0, c, n).astype(np.int32)
    cats = [f"c{i}" for i in range(c)]
    pairs = np.stack([rng.integers(0, n, 30_000), rng.integers(0, n, 30_000)], 1).astype(np.int64)
    lchd = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.15]))
    sess = DeviceSession(lchd)
    a, b = sess.upload(xa, ca), sess.upload(xb, cb)
    anchors = torch.from_numpy(pairs).cuda()
    tag = np.zeros(n, dtype=np.int32)
    want = np.asarray(oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.15])).from_arrays(xa, ca, tag, xb, cb, tag, pairs, 10.0))
    direct = sess.from_primitives(a, b, anchors, 10.0).clone()
    assert np.max(np.abs(direct.cpu().numpy() - want)) < TIGHT
    yield dict(lh=lh, sess=sess, a=a, b=b, anchors=anchors, n=n, want=want, direct=direct, xa=xa, xb=xb, ca=ca, cb=cb, cats=cats, pairs=pairs)
    sess.close()


def test_partition_kernels_equal_the_torch_rule(setup):
    import torch
    from loco_hd_amd.dist import select_shard, shard_rule, unshard

    sess = setup["sess"]
    rng = np.random.default_rng(11)
    lists = {
        "random": (setup["anchors"], setup["n"]),
        "big": (torch.from_numpy(np.stack([rng.integers(0, 200_000, 1_000_003), rng.integers(0, 200_000, 1_000_003)], 1)).cuda(), 200_000),
        # one reference anchor against thousands (python_codes/kras_scan.py:46-52): binned by side B, or -- without the size of
        # structure B / with both sides degenerate -- cut into contiguous slices
        "one_anchor": (torch.from_numpy(np.stack([np.full(5000, 17), np.arange(5000)], 1)).cuda(), 6000, 5000),
        "one_anchor_no_b": (torch.from_numpy(np.stack([np.full(5000, 17), np.arange(5000)], 1)).cuda(), 6000, None),
        "one_pair_repeated": (torch.from_numpy(np.stack([np.full(4097, 3), np.full(4097, 9)], 1)).cuda(), 50, 60),
        "out_of_range": (torch.tensor([[-3, 0], [10**12, 1], [5, 2], [99, 3]], dtype=torch.int64).cuda(), 100, 100),
        "single_pair": (torch.tensor([[7, 7]], dtype=torch.int64).cuda(), 10, 10),
    }
    lists["random"] = lists["random"] + (setup["n"],)
    lists["big"] = lists["big"] + (200_000,)
    for name, (anc, n_atoms, n_atoms_b) in lists.items():
        anc = anc.contiguous()
        p = anc.shape[0]
        for world in (1, 2, 3, 8, 64):
            rank_of_pair, counts = shard_rule(anc, n_atoms, world, n_atoms_b)
            if name.startswith("one_") and world in (2, 3, 8):
                assert max(counts) <= p // world + 64, (name, world, counts)  # balanced, whichever key side the rule fell back to
            stride = max(max(counts), 1)
            gathered = torch.zeros((world, 2, stride), dtype=torch.float64, device="cuda")
            fake = torch.arange(p, dtype=torch.float64, device="cuda") * 0.5 + 1.0  # "score" of pair i
            for r in range(world):
                sel, idx, c2 = select_shard(anc, n_atoms, world, r, session=sess, n_atoms_b=n_atoms_b)
                assert c2 == counts, (name, world, r)
                want_idx = (rank_of_pair == r).nonzero().reshape(-1)
                assert torch.equal(torch.sort(idx).values, want_idx), (name, world, r)  # same SET (the kernel's order is arrival order)
                assert torch.equal(sel, anc[idx])
                gathered[r, 0, : counts[r]] = fake[idx]
                gathered[r, 1, : counts[r]] = idx.view(torch.float64)
            out = unshard(gathered.reshape(-1), counts, stride, p, session=sess)
            assert torch.equal(out, fake), (name, world)


def test_every_rank_of_an_emulated_world_on_one_gpu(setup):
    """What 4 ranks would do, one after the other on this GPU: bitwise the single call, in anchor-pair order."""
    import torch
    from loco_hd_amd.dist import select_shard, unshard

    s = setup
    world = 4
    parts = []
    for r in range(world):
        sel, idx, counts = select_shard(s["anchors"], s["n"], world, r, session=s["sess"])
        parts.append((s["sess"].from_primitives(s["a"], s["b"], sel, 10.0).clone(), idx))
    stride = max(counts)
    gathered = torch.zeros((world, 2, stride), dtype=torch.float64, device="cuda")
    for r, (sc, idx) in enumerate(parts):
        gathered[r, 0, : counts[r]] = sc
        gathered[r, 1, : counts[r]] = idx.view(torch.float64)
    out = unshard(gathered.reshape(-1), counts, stride, s["anchors"].shape[0], session=s["sess"])
    assert torch.equal(out, s["direct"])
    # each rank built about a quarter of side A's environments
    uniq_a = [int(torch.unique(s["anchors"][idx][:, 0]).numel()) for _, idx in parts]
    assert max(uniq_a) < 0.3 * s["n"], uniq_a


def test_score_sharded_under_rccl_world_of_one(setup):
    """DeviceSession + torch.distributed (backend nccl = RCCL): world size 1 with the collective forced, both partitions."""
    import torch
    import torch.distributed as dist
    from loco_hd_amd.dist import score_sharded

    s = setup
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29700 + os.getpid() % 1000)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        fn = lambda sub: s["sess"].from_primitives(s["a"], s["b"], sub, 10.0)
        for partition in ("anchor", "contiguous"):
            full = score_sharded(fn, s["anchors"], 1, 0, n_atoms_a=s["n"], session=s["sess"], partition=partition, force_collective=True)
            torch.cuda.synchronize()
            assert torch.equal(full, s["direct"]), partition
    finally:
        dist.destroy_process_group()


def test_device_group_of_the_c_abi(setup, oracle):
    """lchd_group_from_primitives with two contexts on device 0: the call a Rust / C binding would make for multi-GPU."""
    s, lh = setup, setup["lh"]
    tag = np.zeros(s["n"], dtype=np.int32)
    pa, pb = lh.api._Packed(s["xa"], s["ca"], tag), lh.api._Packed(s["xb"], s["cb"], tag)
    single = lh.LoCoHD(s["cats"], lh.WeightFunction("hyper_exp", [1.0, 0.15]))
    ref = single.from_packed(pa, pb, s["pairs"], 10.0)
    for devices in ([0], [0, 0], [0, 0, 0]):
        grp = lh.LoCoHD(s["cats"], lh.WeightFunction("hyper_exp", [1.0, 0.15]), devices=devices)
        got = grp.from_packed(pa, pb, s["pairs"], 10.0)
        assert np.array_equal(got, ref), devices
        if len(devices) > 1:
            counts = grp.last_group_counts()
            assert sum(counts) == len(s["pairs"]) and min(counts) > 0.25 * len(s["pairs"]) / len(devices), counts
        # errors come back as the reference's classes from whichever device meets them
        bad = s["pairs"][:100].copy()
        bad[37, 1] = s["n"] + 5
        with pytest.raises(lh.PanicException):
            grp.from_packed(pa, pb, bad, 10.0)
        # ... and the group keeps working (a call this small takes the one-launch sweep, one pair per wavefront: its lanes cut a pair
        # into other chunks than the two-pairs-per-wavefront kernel of the large call -- last-bit differences)
        assert np.max(np.abs(grp.from_packed(pa, pb, s["pairs"][:1000], 10.0) - ref[:1000])) < 1e-13
    assert np.max(np.abs(ref - s["want"])) < TIGHT
    # the reference's call shape with lists of PrimitiveAtoms and a dictionary of weight functions (per-pair keys travel too)
    multi = {"near": lh.WeightFunction("uniform", [0.0, 6.0]), "far": lh.WeightFunction("hyper_exp", [1.0, 0.1])}
    omulti = {"near": oracle.WeightFunction("uniform", [0.0, 6.0]), "far": oracle.WeightFunction("hyper_exp", [1.0, 0.1])}
    prims = lambda mod, x, c: [mod.PrimitiveAtom(s["cats"][k], "", xyz) for k, xyz in zip(c[:600], x[:600])]
    keyed = [(int(i) % 600, int(j) % 600, "near" if k % 3 else "far") for k, (i, j) in enumerate(s["pairs"][:900])]
    got = lh.LoCoHD(s["cats"], multi, devices=[0, 0]).from_primitives(prims(lh, s["xa"], s["ca"]), prims(lh, s["xb"], s["cb"]), keyed, 9.0)
    want = oracle.LoCoHD(s["cats"], omulti).from_primitives(prims(oracle, s["xa"], s["ca"]), prims(oracle, s["xb"], s["cb"]), keyed, 9.0)
    assert np.max(np.abs(np.asarray(got) - np.asarray(want))) < TIGHT


@pytest.mark.parametrize("workload,extra", [("c2a", ["--pairs", "200000"]), ("c5", ["--pairs", "200000"]), ("c3", [])])
def test_bench_strong_scaling_line_under_a_forced_process_group(workload, extra):
    """The first real multi-GPU run of bench.py must not die on plumbing: `--scaling strong` with the RCCL process group, the
    per-step asynchronous gather and the restore of anchor-pair order forced on ONE rank (LCHD_BENCH_FORCE_DIST=1).  The JSON line
    carries the collective, the roofline block and the plan-cache note; bench.py itself asserts that the gathered + restored
    scores equal a single pass over the whole list (src/locohd.rs:545-557: results in pair order)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, LCHD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 150),
               WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cmd = [sys.executable, str(root / "bench.py"), "--workload", workload, "--scaling", "strong", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", *extra]
    run = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert "RCCL gather" in line["config"]["collective"]
    for key in ("bound", "achieved", "peak", "frac", "unit"):
        assert key in line["roofline"], key
    assert 0 < line["roofline"]["frac"] < 1.5
    if workload != "c3":
        assert line["config"]["plan_cache"] is True and line["config"]["plan_ms"] > 0


def _bench_without_launcher(extra_env, *argv, timeout=900):
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LCHD_BENCH_FORCE_DIST")}
    env.update(extra_env)
    return subprocess.run([sys.executable, str(root / "bench.py"), *argv], cwd=root, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("workload,extra", [("c2a", ["--pairs", "200000"]), ("c3", [])])
def test_bench_with_two_real_ranks_started_by_bench_itself(workload, extra):
    """`python3 bench.py --gpus 2` -- the driver's command form, no launcher -- with TWO real ranks on this box's one GPU
    (LCHD_BENCH_SHARE_GPU: gloo + host-staged gather instead of RCCL, which refuses two ranks on one device): the ranks start, both
    partition the one list with the library's kernels, score their shares, rank 0 gathers, restores anchor-pair order and bench.py
    asserts the result equals a single pass over the whole list (src/locohd.rs:545-557).  Everything of an N-rank run but the
    RCCL transport."""
    import json

    run = _bench_without_launcher({"LCHD_BENCH_SHARE_GPU": "1"}, "--gpus", "2", "--workload", workload, "--steps", "2", "--warmup", "1",
                                  "--no-cpu-baseline", *extra)
    assert run.returncode == 0, run.stderr[-3000:]
    assert "rank 0 of 2 up" in run.stderr and "rank 1 of 2 up" in run.stderr
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and "rehearsal" in line
    assert "gather" in line["config"]["collective"]
    if workload == "c2a":
        assert 0.3 * 200000 < line["config"]["pairs_this_rank"] < 0.7 * 200000  # rank 0 scored its share, not the list


def test_bench_gpus_2_on_one_gpu_fails_loudly():
    """The driver's exact command on a box with fewer GPUs than ranks: a clear message and a non-zero exit code, no hang."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the command would run the real benchmark")
    run = _bench_without_launcher({}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=300)
    assert run.returncode != 0
    assert "needs 2 visible devices" in run.stderr, run.stderr[-2000:]
