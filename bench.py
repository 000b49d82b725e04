#!/usr/bin/env python3
"""bench.py -- anchor-pair LoCoHD scores/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workloads (--workload):
  c2a (default; BASELINE.json configs[1])  two 10 000-atom random clouds at protein-like density 0.05 atoms/A^3 (box side
        58.5 A), 10 primitive categories, hyper_exp[1.0, 0.1], Hellinger-2, from_primitives semantics with threshold 10 A,
        P = 10^6 anchor pairs: pair k = (k mod N, perm_r(k mod N)), r = k div N, perm_r a seeded permutation (round 0 =
        identity), so every one of the 10^4 atoms of each cloud is an anchor ~100 times.
  c2b   the same two clouds, dense from_coords semantics: 10^4 pairs (i, i), every environment is the whole structure.
  c3    (configs[2]) 50 decoys x 3000 atoms, 8 categories, every 3rd atom a "Cent" anchor, tag rule accept_same=False,
        uniform[3,10], all 1225 unordered decoy pairs = 1.225 x 10^6 anchor pairs in ONE batched call.
  c4    (configs[3]) one reference vs 5000 trajectory frames of a 2001-primitive-atom system given as float32 source atoms.
  c5    (configs[4]) two 200 000-point clouds, 25 categories, 10^6 random anchor pairs.
Coordinates, categories and anchor pairs are resident in HBM before the timed region; ONE STEP = one full pass of the hot
path over the batch: cell lists for both clouds, anchor de-duplication, environment build (radius search + f64 distances +
sort), merge sweep (Hellinger + CDF + reduction), status hand-over.  Nothing is cached across steps.
Order of a run: set-up (contexts, uploads, PRIME_STEPS = 6 untimed full passes: the GPU leaves its idle clocks -- the first ~4
passes after start-up run ~6 % slow -- and every context has seen the workload once: a pass picks its sweep kernels from the
previous pass's pair statistics), then the W warm-up steps, barrier + synchronize, EXACTLY K timed steps, barrier + synchronize.

Multi-GPU (--scaling):
  strong (default for N > 1 ranks -- WORLD_SIZE or --gpus --, and for --emulate-world): ONE list of P pairs for the whole job
        (BASELINE configs[2] and [4] are fixed-size 8-GPU jobs): every rank partitions the full list with the library's kernels
        (pairs binned by their side-A anchor, loco_hd_amd/dist.py; planned once per list and reused while the tensor is unchanged,
        --no-plan-cache plans in every step), scores its share, and the shares travel to rank 0, which restores anchor-pair order.
  weak (default at N = 1; anchor pairs are independent, the path has no exchange step): every rank holds both structures and its
        OWN P pairs; value = N x P pairs per max-over-ranks step time.  c3 shards whole decoy pairs (tiles of the
        50 x 50 pair matrix) so that a rank touches few decoys.
  In both modes the RCCL gather of the scores to rank 0 is INSIDE the timed steps for N > 1 (asynchronous, overlapped with the
  next step's scoring; `--gather end` moves it behind the timed region).
  --emulate-world W (one GPU): the W ranks' strong-scaling steps one after the other on this GPU, no collective: per-rank step
  times for the scaling estimate in DESIGN.md section 6.

The JSON line also carries:
  roofline      dominant kernel's ALGORITHMIC bytes per launch (B_pair = (n_A+n_B)*28 + 16) / its average
                launch duration (HIP events on the launch stream, recorded by the C library) against the 8 TB/s HBM peak
                (`frac`, as section 8d defines it), plus what actually binds the kernel: `fabric_measured_frac` (fabric-side
                bytes of the committed rocprofv3 PMC passes / the PROFILED duration / peak; Infinity-Cache hits counted) and `valu_issue_frac` (SQ_ACTIVE_INST_VALU
                x 4 cycles / 1024 SIMDs / the profiled duration at the counter run's clock).
  cpu_baseline  the CPU oracle (C restatement of the reference algorithm, kind "port": the Rust reference cannot be built
                here) timed on this host's cores on a bounded sample of the same pairs; the same sample is the parity gate
                (max |gpu - cpu| must be <= 1e-6).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
PROFILE_ROUND = "r06"
PRIME_STEPS = 6  # untimed set-up passes in front of the warm-up steps (Harness.run)


# ---------------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------------
def make_workload(name: str, rank: int, n_pairs: int, same_on_all_ranks: bool = False, n_atoms: int = 10_000):
    """Synthetic pair-list inputs of SURVEY.md 8(d).  Returns dict(xyz_a, xyz_b, cat_a, cat_b, pairs, thr, C, wf, label)."""
    if name in ("c2a", "c2b"):
        n, c, seed, thr = (n_atoms if name == "c2b" else 10_000), 10, 2, 10.0
        rounds = max(1, n_pairs // n)
    elif name == "c5":  # stress: two 200k-point clouds, 25 categories, random anchor pairs
        n, c, seed, thr = 200_000, 25, 5, 10.0
        rounds = 0
    else:
        raise SystemExit(f"unknown workload {name}")
    rng = np.random.default_rng(seed)
    side = (n / 0.05) ** (1.0 / 3.0)
    xyz_a, xyz_b = rng.uniform(0.0, side, (n, 3)), rng.uniform(0.0, side, (n, 3))
    cat_a, cat_b = rng.integers(0, c, n).astype(np.int32), rng.integers(0, c, n).astype(np.int32)
    prank = 0 if same_on_all_ranks else rank
    prng = np.random.default_rng(1000 * seed + prank)  # weak scaling: every rank scores different pairs of the same clouds
    if name == "c2b":
        idx = np.arange(n, dtype=np.int64)
        pairs = np.stack([idx, idx], 1)
        label = f"C2b: 2x{n}-atom random clouds (0.05 atoms/A^3), {c} categories, hyper_exp[1,0.1], Hellinger-2, dense from_coords: {n} pairs (i, i), environment = whole structure"
    elif rounds:
        idx = np.arange(n, dtype=np.int64)
        cols = []
        for r in range(rounds):
            perm = idx if (r == 0 and prank == 0) else prng.permutation(n)
            cols.append(np.stack([idx, perm], 1))
        pairs = np.concatenate(cols, 0)[:n_pairs]
        label = (f"C2a: 2x{n}-atom random clouds (0.05 atoms/A^3), {c} categories, hyper_exp[1,0.1], Hellinger-2, "
                 f"from_primitives thr {thr:g} A, {len(pairs)} anchor pairs = {rounds} permutation rounds over all atoms")
    else:
        pairs = np.stack([prng.integers(0, n, n_pairs), prng.integers(0, n, n_pairs)], 1).astype(np.int64)
        label = f"C5: 2x{n}-point clouds, {c} categories, hyper_exp[1,0.1], {n_pairs} random anchor pairs, thr {thr:g} A"
    return dict(xyz_a=xyz_a, xyz_b=xyz_b, cat_a=cat_a, cat_b=cat_b, pairs=np.ascontiguousarray(pairs), thr=thr, C=c,
                wf=("hyper_exp", [1.0, 0.1]), label=label, n=n)


def make_c3(rank: int, same_on_all_ranks: bool):
    """BASELINE configs[2]: 50 decoys x 3000 points (every 3rd a "Cent" anchor of a 3-atom residue), all unordered decoy pairs."""
    rng = np.random.default_rng(3 + (0 if same_on_all_ranks else 1000 * rank))
    types = ["Cent", "AmideC", "OH", "Pos", "Neg", "Aro", "Ali", "Sulf"]
    n, nd = 3000, 50
    side = (n / 0.023) ** (1 / 3)
    base = rng.uniform(0, side, (n, 3))
    cat = np.where(np.arange(n) % 3 == 0, 0, rng.integers(1, 8, n)).astype(np.int32)
    tag = (np.arange(n) // 3).astype(np.int32)
    decoys = [(base + rng.normal(0, 1.5, base.shape), cat, tag) for _ in range(nd)]
    spairs = [(a, b) for a in range(nd) for b in range(a + 1, nd)]
    label = (f"C3: {nd} decoys x {n} atoms (0.023 atoms/A^3), 8 categories, {n // 3} 'Cent' anchors per decoy, accept_same=False, "
             f"uniform[3,10], thr 10 A, all {len(spairs)} unordered decoy pairs in one batched call")
    return dict(types=types, decoys=decoys, spairs=spairs, n=n, nd=nd, label=label, thr=10.0)


def tile_structure_pairs(nd: int, world: int):
    """Strong-scaling partition of the upper triangle of the nd x nd decoy-pair matrix into `world` shares of whole decoy pairs.
    A rank builds the environments of every decoy its pairs touch, so the shares are made of square TILES of the pair matrix
    (a tile of e x e pairs touches 2e decoys; e^2 pairs in a row of the matrix would touch e^2 + 1).
    world = g^2 / 2 for an even g (2, 8, 18, 32): the decoys are cut into g groups; each of the g (g - 1) / 2 off-diagonal tiles is
    one share, and the g diagonal tiles (half as many pairs each) go two to a share -- every rank touches 2 groups = 2 nd / g
    decoys (world 8: 25 of 50) and the shares are equal to within a group's rounding.
    Otherwise: tiles of about half a share, dealt largest first to the rank with the fewest pairs."""
    g = 2
    while g * g // 2 < world:
        g += 2
    if g * g // 2 == world:
        bounds = [round(k * nd / g) for k in range(g + 1)]
        groups = [range(bounds[k], bounds[k + 1]) for k in range(g)]
        shares = [[(a, b) for a in groups[i] for b in groups[j]] for i in range(g) for j in range(i + 1, g)]
        diag = [[(a, b) for a in groups[i] for b in groups[i] if a < b] for i in range(g)]
        shares += [diag[2 * k] + diag[2 * k + 1] for k in range(g // 2)]
        return shares
    g = 1
    while g * (g + 1) // 2 < 2 * world:
        g += 1
    edge = -(-nd // g)
    tiles = []
    for i in range(g):
        for j in range(i, g):
            ps = [(a, b) for a in range(i * edge, min((i + 1) * edge, nd)) for b in range(j * edge, min((j + 1) * edge, nd)) if a < b]
            if ps:
                tiles.append(ps)
    tiles.sort(key=len, reverse=True)
    shares = [[] for _ in range(world)]
    for t in tiles:
        min(shares, key=len).extend(t)
    return shares


def profile_numbers(workload: str):
    """What the committed rocprofv3 passes of this workload say about its dominant kernel (profiles/<round>/traffic_<w>.json,
    written by profiles/make_traffic.py from separate --pmc passes): PMC counters cannot be read from inside this process."""
    try:
        return json.loads((ROOT / "profiles" / PROFILE_ROUND / f"traffic_{workload}.json").read_text())
    except (OSError, ValueError):
        return None


def roofline_block(workload: str, kernel: str, algo_bytes: float, launch_ms: float, launches_per_step: int = 1, step_ms: float = 0.0):
    achieved = algo_bytes / max(launch_ms, 1e-9) / 1e6  # GB/s
    block = {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "traffic": None, "kernel": kernel, "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": launch_ms,
             "launches_per_step": launches_per_step,
             "note": "achieved / peak / frac = ALGORITHMIC bytes (SURVEY.md 8d) of the dominant kernel / its live launch time against the "
                     "8 TB/s HBM peak, as section 8d defines the figure; frac_step = the algorithmic bytes of the WHOLE step / ms_per_step / "
                     "peak (what the byte model actually covers: points in, score out).  What binds the kernels of this path is VALU "
                     "issue, not bandwidth (`bound`): valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / kernel cycles, "
                     "fabric_measured_frac = PMC traffic / the profiled launch time / peak (Infinity-Cache hits counted); frac_no_reuse "
                     "(c2a) = the same byte model over the unique-anchor call, where every environment is built for ONE pair; valu_ceiling_frac = (vector instructions of the launch per SIMD x the "
                     "class-weighted cost of the kernel's hot-loop instruction mix, measured per class by profiles/ubench/issue_rates.hip) / "
                     "launch time: 1.0 = the kernel issues as fast as its own instruction mix allows"}
    if step_ms > 0.0:
        block["frac_step"] = algo_bytes * launches_per_step / (step_ms * 1e-3) / (HBM_PEAK_GBS * 1e9)
    prof = profile_numbers(workload)
    if prof and prof.get("kernel") and prof["kernel"].split("<")[0] in kernel:
        block["traffic"] = prof.get("traffic_bytes_per_launch")
        block["traffic_source"] = f"profiles/{PROFILE_ROUND}/traffic_{workload}.json (separate rocprofv3 --pmc passes of this command)"
        prof_ns = prof.get("avg_launch_ns_kernel_trace")
        if block["traffic"] and prof_ns:
            # counter bytes over the PROFILED launch time of the same collection (not this run's live time: another box, another
            # clock).  FETCH_SIZE counts requests at the fabric, Infinity-Cache (MALL) hits included (MI355X_MICROARCH.md, HBM
            # section): an upper bound of the HBM traffic, hence "fabric", not "hbm"
            block["fabric_measured_frac"] = block["traffic"] / (prof_ns * 1e-9) / (HBM_PEAK_GBS * 1e9)
            block["fabric_measured_note"] = ("traffic / avg_launch_ns_kernel_trace of the same profile / 8 TB/s; the counters see fabric "
                                             "requests, Infinity-Cache hits included: an upper bound of what reached HBM")
            block["profiled_launch_ms"] = prof_ns * 1e-6
        if prof.get("valu_issue_frac") is not None:
            block["valu_issue_frac"] = prof["valu_issue_frac"]
        for k in ("valu_ceiling_frac", "valu_avg_ns_per_instr_measured", "valu_avg_ns_per_instr_class_mix", "valu_share_of_2_cycle_class",
                  "valu_ceiling_source"):
            if prof.get(k) is not None:
                block[k] = prof[k]
        block["binding"] = prof.get("binding", "valu issue")
    return block


def usable_cores() -> int:
    """CPU threads this process may really use: affinity mask and cgroup quota, not just os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(w, gpu_scores: np.ndarray, budget_s: float = 12.0):
    """Time the CPU oracle on a bounded prefix of the same pairs (all host cores) and use it as the parity gate."""
    from oracle import oracle as orc  # checker / baseline only

    cores = usable_cores()
    lchd = orc.LoCoHD([f"c{i}" for i in range(w["C"])], orc.WeightFunction(*w["wf"]), n_of_threads=cores)
    tag = np.zeros(w["n"], dtype=np.int32)

    def run(m):
        t0 = time.perf_counter()
        out = lchd.from_arrays(w["xyz_a"], w["cat_a"], tag, w["xyz_b"], w["cat_b"], tag, w["pairs"][:m], w["thr"])
        return np.asarray(out), time.perf_counter() - t0

    probe = min(len(w["pairs"]), 500 * cores)
    _, dt = run(probe)
    rate = probe / max(dt, 1e-9)
    m = int(min(len(w["pairs"]), max(10_000, rate * budget_s)))
    ref, dt = run(m)
    err = float(np.max(np.abs(ref - gpu_scores[:m])))
    one = orc.LoCoHD([f"c{i}" for i in range(w["C"])], orc.WeightFunction(*w["wf"]), n_of_threads=1)
    m1 = int(min(len(w["pairs"]), max(2_000, rate / cores * 3.0)))  # ~3 s on one core
    t0 = time.perf_counter()
    one.from_arrays(w["xyz_a"], w["cat_a"], tag, w["xyz_b"], w["cat_b"], tag, w["pairs"][:m1], w["thr"])
    dt1 = time.perf_counter() - t0
    return {"value": m / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "value_1thread": m1 / dt1,
            "sample": f"first {m} anchor pairs of the same workload, {dt:.1f} s, C oracle (oracle/locohd_oracle.c) with "
                      f"{cores} pthreads over anchor pairs; reference Rust core not buildable in this image"}, err, m


class Harness:
    """warm-up, barrier + synchronize, K timed steps, drain, barrier + synchronize, max over ranks"""

    def __init__(self, args, torch, dist, dev, use_dist):
        self.args, self.torch, self.dist, self.dev, self.use_dist = args, torch, dist, dev, use_dist

    def run(self, step, drain=lambda: None, on_timed_start=lambda: None):
        a, torch, dist = self.args, self.torch, self.dist
        # set-up, before the W warm-up steps the caller asked for: PRIME_STEPS passes that bring the GPU out of its idle power
        # state (the first ~4 steps after start-up run ~6 % slower: clock ramp) and let every session see one pass of its
        # workload (the launch set of a pass is picked from the previous pass's pair statistics); untimed, full work
        for _ in range(PRIME_STEPS):
            step()
        drain()
        for _ in range(a.warmup):
            step()
        drain()
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        on_timed_start()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
            if os.environ.get("LCHD_BENCH_DEBUG"):
                print("step done at", (time.perf_counter() - t0) * 1e3, "ms", file=sys.stderr)
        drain()  # every pass and every collective of the timed steps has completed inside the timed region
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if self.use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed


def base_result(args, world, total_pairs, elapsed, label, extra_cfg):
    return {
        "metric": "anchor-pair LoCoHD scores/sec", "value": total_pairs * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": label, "setup_passes_before_warmup": PRIME_STEPS, "streams": getattr(args, "streams", 1),
                   "streams_note": ("consecutive steps alternate between two sessions; on ONE stream (the default at N = 1) they run one after the other, "
                                    "so a launch of the dominant kernel has the chip to itself and roofline.achieved / frac mean what SURVEY 8d defines; "
                                    "on TWO streams (the default for N > 1 and --emulate-world, --streams 2) the steps overlap -- measured at N = 1: "
                                    "+9 % c2a, +10 % c3, +3 % c5 (profiles/r06/x_two_streams_at_n1.txt) -- and a like-for-like single-GPU reference "
                                    "of an N > 1 line is `--gpus 1 --streams 2`"),
                   **extra_cfg},
    }


def collective_note(args, world, use_dist):
    if not use_dist:
        return {}
    if args.gather == "end":
        return {"collective": "none in the timed steps (independent anchor pairs: no exchange step); one RCCL gather to rank 0 after them"}
    what = "scores + original pair positions (16 B per pair)" if args.scaling == "strong" else "scores (8 B per pair)"
    return {"collective": f"RCCL gather of every rank's {what} to rank 0 in every timed step, asynchronous (overlaps the next step's scoring)"
                          + ("; rank 0 restores anchor-pair order" if args.scaling == "strong" else "")}


# ---------------------------------------------------------------------------------------------------------------------------
# C4: trajectory mode
# ---------------------------------------------------------------------------------------------------------------------------
def run_c4(args, torch, dist, dev, rank, world, use_dist):
    """BASELINE.json configs[3] (MD-trajectory mode) with the structure -> primitive-atom step on the device as well:
    one reference structure vs F frames of the same system, per-residue "Cent" anchors, accept_same=False, uniform[3,10],
    threshold 10 A (python_codes/trajectory_analyzer.py:76-119).  The frames are float32 SOURCE atoms resident in HBM
    ([F][5336][3]); one step = for every chunk of frames: k_frames_centroids (2001 primitive atoms per frame from their
    source atoms, np.mean arithmetic) + the full scoring pass (cell lists, environments, sweep) of the chunk's anchor pairs.
    Frames shard across ranks with no exchange step: weak = F frames per rank, strong = F frames in total."""
    import loco_hd_amd as lh
    from loco_hd_amd.device import DeviceSession

    n_res, per_res, chunk = 667, 8, args.chunk
    n_frames = args.frames if args.scaling == "weak" else max(1, args.frames // world)
    n_src, n_prim = n_res * per_res, 3 * n_res
    rng = np.random.default_rng(4 + 1000 * rank)
    side = (n_prim / 0.023) ** (1.0 / 3.0)
    centres = rng.uniform(0.0, side, (n_res, 3))
    ref_atoms = (np.repeat(centres, per_res, 0) + rng.normal(0.0, 1.5, (n_src, 3))).astype(np.float32)
    # CSR map: primitive 3r = centroid of the residue's 8 atoms, 3r+1 = atom 1, 3r+2 = centroid of atoms 4 and 5
    src_start, src_idx = [0], []
    for r in range(n_res):
        for members in (range(per_res), (1,), (4, 5)):
            src_idx += [per_res * r + m for m in members]
            src_start.append(len(src_idx))
    types = ["Cent", "AmideC", "OH", "Pos", "Neg", "Aro", "Ali", "Sulf"]
    cat = np.zeros(n_prim, dtype=np.int32)
    cat[1::3], cat[2::3] = rng.integers(1, 8, n_res), rng.integers(1, 8, n_res)
    tag = np.repeat(np.arange(n_res, dtype=np.int32), 3)

    class Topo:  # the fields DeviceSession.set_frame_sources reads (PrimitiveTopology of a synthetic structure)
        pass
    topo = Topo()
    topo.src_start, topo.src_idx, topo.n_atoms = np.asarray(src_start, np.int32), np.asarray(src_idx, np.int32), n_src

    def centroids(atoms):  # np.mean per primitive atom, float32 (what the reference computes per frame on the host)
        return np.stack([np.mean(atoms[topo.src_idx[a:b]], axis=0) for a, b in zip(src_start[:-1], src_start[1:])]).astype(np.float64)

    lchd = lh.LoCoHD(types, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule({"accept_same": False}))
    sess = DeviceSession(lchd, device=dev.index)
    sess.enable_timing(True)
    ref = sess.upload(centroids(ref_atoms), cat, tag)
    # --c4-sessions 2 (default): the chunks of a step alternate between TWO sessions (contexts) with a frames buffer each, enqueued
    # asynchronously (from_primitives_async; a session is waited for when its turn comes again): chunk k + 1's conversion runs on a
    # side stream while chunk k is scored, so the host's grid planning (it needs the chunk's bounding box) does not wait for the
    # scoring stream and the device never idles between chunks.  1: one session, every chunk waited for before the next is enqueued.
    n_sess = max(1, min(2, args.c4_sessions))
    sessions = [sess] + [DeviceSession(lchd, device=dev.index) for _ in range(n_sess - 1)]
    for s_ in sessions[1:]:
        s_.enable_timing(True)
    refs = [ref] + [s_.upload(centroids(ref_atoms), cat, tag) for s_ in sessions[1:]]
    side_stream = torch.cuda.Stream(device=dev) if n_sess > 1 else None
    gen = torch.Generator(device=dev).manual_seed(4 + rank)
    d_ref = torch.from_numpy(ref_atoms).to(dev)
    frames = (d_ref[None] + 0.5 * torch.randn((n_frames, n_src, 3), generator=gen, device=dev, dtype=torch.float32)).contiguous()
    chunk = min(chunk, n_frames)
    buf = sess.frames_buffer(ref, chunk)
    sess.set_frame_sources(buf, topo)
    bufs = [buf]
    for s_, r_ in zip(sessions[1:], refs[1:]):
        b_ = s_.frames_buffer(r_, chunk)
        s_.set_frame_sources(b_, topo)
        bufs.append(b_)
    la = torch.arange(0, n_prim, 3, dtype=torch.int64, device=dev)
    offs = torch.arange(chunk, dtype=torch.int64, device=dev).repeat_interleave(len(la)) * n_prim
    anchors = torch.stack([la.repeat(chunk), la.repeat(chunk) + offs], 1).contiguous()
    p = n_frames * len(la)
    out = torch.empty(p, dtype=torch.float64, device=dev)
    starts = list(range(0, n_frames, chunk))
    phase = {"convert": 0.0, "cells": 0.0, "anchors": 0.0, "env": 0.0, "sweep": 0.0}
    collect = [False]
    env_points = [0]

    busy = [False] * n_sess

    def drain(i):  # wait for session i's chunk (if one is in flight) and book its phase times
        if not busy[i]:
            return
        sessions[i].finish()
        busy[i] = False
        if collect[0]:
            phase["convert"] += sessions[i].last_convert_ms(bufs[i])
            for k, v in sessions[i].last_ms().items():
                phase[k] += v

    def step():
        if n_sess == 1:
            for f0 in starts:
                nf = min(chunk, n_frames - f0)
                sess.load_atom_frames_dev(buf, frames[f0:f0 + nf])
                sess.from_primitives(ref, buf, anchors[: nf * len(la)], 10.0, out=out[f0 * len(la):(f0 + nf) * len(la)])
                if collect[0]:
                    phase["convert"] += sess.last_convert_ms(buf)
                    for k, v in sess.last_ms().items():
                        phase[k] += v
            return
        for c, f0 in enumerate(starts):
            i = c % n_sess
            nf = min(chunk, n_frames - f0)
            drain(i)
            sessions[i].load_atom_frames_dev(bufs[i], frames[f0:f0 + nf], side_stream)
            sessions[i].from_primitives_async(refs[i], bufs[i], anchors[: nf * len(la)], 10.0, out[f0 * len(la):(f0 + nf) * len(la)])
            busy[i] = True
        for i in range(n_sess):  # (a step ends with every chunk scored)
            drain(i)

    def start_collect():
        collect[0] = True

    elapsed = Harness(args, torch, dist, dev, use_dist).run(step, on_timed_start=start_collect)
    collect[0] = False
    phase = {k: v / max(args.steps, 1) for k, v in phase.items()}
    for f0 in starts:  # environment points of the whole job (outside the timed region)
        nf = min(chunk, n_frames - f0)
        sess.load_atom_frames_dev(buf, frames[f0:f0 + nf])
        sess.from_primitives(ref, buf, anchors[: nf * len(la)], 10.0, out=out[f0 * len(la):(f0 + nf) * len(la)])
        env_points[0] += sess.last_env_points()
    scores = out.cpu().numpy().reshape(n_frames, len(la))
    result = None
    if rank == 0:
        algo = env_points[0] * 32 + 16 * p  # 32 B per environment point when a tag rule is active
        dom = max(("env", "sweep"), key=lambda k: phase[k])
        conv_bytes = n_frames * (n_src * 12 + n_prim * 24)  # every source atom read once (3 x f32), every primitive atom written (3 x f64)
        label = (f"C4: 1 reference vs {n_frames} frames/GPU of a {n_prim}-primitive-atom system ({n_src} float32 source atoms/frame, "
                 f"converted on the device), {len(la)} per-residue anchors/frame, accept_same=False, uniform[3,10], thr 10 A, "
                 f"chunks of {chunk} frames")
        result = base_result(args, world, p * world, elapsed, label, {"pairs_per_gpu": p, "mean_env_points_per_pair": env_points[0] / p,
                                                                      "sharding": "frames across ranks, no exchange step, scores stay on their rank",
                                                                      "sessions": n_sess})
        result["roofline"] = roofline_block("c4" if dom == "sweep" else "c4_env",
                                            {"env": "k_env_group<false, 320, 6> (four environments per wavefront; side B: one environment per pair, no de-duplication)",
                                             "sweep": "k_sweep_duo<8, 16 lanes, 240 events> (four pairs per wavefront)"}[dom], algo / len(starts),
                                            phase[dom] / len(starts), len(starts), step_ms=result["ms_per_step"])
        result["kernel_ms"] = phase
        result["extras"] = {"k_frames_centroids": {"ms_per_step": phase["convert"], "algorithmic_bytes_per_step": conv_bytes,
                                                   "GB_per_s": conv_bytes / max(phase["convert"], 1e-9) / 1e6,
                                                   "frac_of_hbm_peak": conv_bytes / max(phase["convert"], 1e-9) / 1e6 / HBM_PEAK_GBS}}
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as orc  # checker / baseline only

            cores = usable_cores()
            lo = orc.LoCoHD(types, orc.WeightFunction("uniform", [3.0, 10.0]), orc.TagPairingRule({"accept_same": False}), n_of_threads=cores)
            sample = sorted(set(np.linspace(0, n_frames - 1, min(n_frames, 24)).astype(int).tolist()))
            host_frames = frames[torch.tensor(sample, device=dev)].cpu().numpy()
            ref_xyz = centroids(ref_atoms)
            pairs = np.stack([np.arange(0, n_prim, 3), np.arange(0, n_prim, 3)], 1).astype(np.int64)
            t1 = time.perf_counter()
            err = 0.0
            for k, f in enumerate(sample):  # per frame, like the reference's loop: convert, then score
                got = np.asarray(lo.from_arrays(ref_xyz, cat, tag, centroids(host_frames[k]), cat, tag, pairs, 10.0))
                err = max(err, float(np.max(np.abs(got - scores[f]))))
            dt = time.perf_counter() - t1
            result["cpu_baseline"] = {"value": len(sample) * len(pairs) / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
                                      "sample": f"{len(sample)} frames spread over the trajectory ({len(sample) * len(pairs)} pairs), {dt:.1f} s: "
                                                f"NumPy np.mean conversion per frame + C oracle with {cores} pthreads over anchor pairs"}
            result["max_abs_err_vs_cpu"] = err
            result["parity_sample_pairs"] = len(sample) * len(pairs)
            if not (err <= 1e-6):
                result["parity_failed"] = True
    for s_ in sessions[1:]:
        s_.close()
    sess.close()
    return result


# ---------------------------------------------------------------------------------------------------------------------------
# C3: all-vs-all decoys, one batched call
# ---------------------------------------------------------------------------------------------------------------------------
def run_c3(args, torch, dist, dev, rank, world, use_dist):
    import loco_hd_amd as lh
    from loco_hd_amd.device import DeviceSession
    from loco_hd_amd.dist import gather_scores

    strong = args.scaling == "strong"
    w = make_c3(rank, same_on_all_ranks=strong)
    lchd = lh.LoCoHD(w["types"], lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule({"accept_same": False}))
    # two sessions take the steps in turn, on two streams with --streams 2 (see run_pairs)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)] if args.streams == 2 else [torch.cuda.current_stream(dev)] * 2
    sessions = []
    for st_ in streams:
        with torch.cuda.stream(st_):
            sessions.append(DeviceSession(lchd, device=dev.index))
    sess = sessions[0]
    for s_ in sessions:
        s_.enable_timing(True)
    la = np.arange(0, w["n"], 3)
    emu = args.emulate_world if (strong and args.emulate_world > 1 and world == 1) else 0
    shares = tile_structure_pairs(w["nd"], emu or world) if strong else [w["spairs"]]
    out_holder, phases_holder = {}, {}

    def run_share(my_spairs, tag):
        # A rank holds only the decoys its share of decoy pairs touches (strong scaling: 25 of the 50 at 8 ranks), as one batch
        # per session; `local` maps a decoy to its position in that batch.
        touched = sorted({d for ab in my_spairs for d in ab})
        local = {d: i for i, d in enumerate(touched)}
        batches, offs = [], None
        for s_ in sessions:
            if touched:
                b_, offs = s_.upload_batch([w["decoys"][d] for d in touched])
                batches.append(b_)
        pairs = (np.concatenate([np.stack([offs[local[a]] + la, offs[local[b]] + la], 1) for a, b in my_spairs])
                 if my_spairs else np.zeros((0, 2), np.int64))
        anchors = torch.from_numpy(np.ascontiguousarray(pairs)).to(dev)
        p = anchors.shape[0]
        outs = [torch.empty(max(p, 1), dtype=torch.float64, device=dev) for _ in range(2)]
        # strong: every rank's share has its own size; the gather slot is the largest share's
        slot = max(len(s) for s in shares) * len(la) if strong else p
        pads = [torch.zeros(slot, dtype=torch.float64, device=dev) for _ in range(2)]
        gathered = [torch.empty(slot * world, dtype=torch.float64, device=dev) if (use_dist and rank == 0) else None for _ in range(2)]
        pending, in_flight, counter = [None, None], [False, False], [0]
        phase_ms = {"cells": 0.0, "anchors": 0.0, "env": 0.0, "sweep": 0.0}
        collect = [False]
        torch.cuda.synchronize()

        def finish(k):
            if in_flight[k]:
                sessions[k].finish()
                in_flight[k] = False
                if collect[0]:
                    for name, v in sessions[k].last_ms().items():
                        phase_ms[name] += v

        def step():
            k = counter[0] % 2
            counter[0] += 1
            with torch.cuda.stream(streams[k]):
                finish(k)
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None
                if p:
                    sessions[k].from_primitives_async(batches[k], batches[k], anchors, w["thr"], outs[k])
                    in_flight[k] = True
                if use_dist and args.gather == "step":
                    pads[k][:p].copy_(outs[k][:p])
                    pending[k] = gather_scores(pads[k], gathered[k], world, rank, force_collective=True, async_op=True)

        def drain():
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    finish(k)
                    if pending[k] is not None:
                        pending[k].wait()
                        pending[k] = None

        def start_collect():
            collect[0] = True

        elapsed = Harness(args, torch, dist, dev, use_dist).run(step, drain, start_collect)
        collect[0] = False
        out_holder[tag] = (outs[(counter[0] - 1) % 2][:p], pairs, p, batches[0] if batches else None, len(touched))
        phases_holder[tag] = {k: v / max(args.steps, 1) for k, v in phase_ms.items()}
        return elapsed

    emulated = None
    if emu:
        per_rank = []
        for r in range(emu):
            per_rank.append(run_share(shares[r], f"r{r}") / args.steps * 1e3)
        emulated = {"world": emu, "per_rank_ms": per_rank, "per_rank_decoy_pairs": [len(s) for s in shares],
                    "per_rank_kernel_ms": [phases_holder[f"r{r}"] for r in range(emu)]}
        shares = [w["spairs"]]
        strong = False  # the reference step of the estimate: the whole job on this GPU
    my = shares[rank] if strong else w["spairs"]
    elapsed = run_share(my, "main")
    out, pairs, p, batch, n_touched = out_holder["main"]
    phase_ms = phases_holder["main"]
    total_pairs = len(w["spairs"]) * len(la) * (1 if args.scaling == "strong" else world)
    result = None
    if rank == 0:
        if p:
            sess.from_primitives(batch, batch, torch.from_numpy(np.ascontiguousarray(pairs)).to(dev), w["thr"], out=out)
        env_points = sess.last_env_points() if p else 0
        algo = env_points * 32 + 16 * p
        dom = max(("env", "sweep"), key=lambda k: phase_ms[k])
        result = base_result(args, world, total_pairs, elapsed, w["label"],
                             {"pairs_this_rank": p, "mean_env_points_per_pair": env_points / max(p, 1),
                              "sharding": (f"tiles of the 50 x 50 decoy-pair matrix, one set per rank; a rank holds the decoys its tiles touch ({n_touched} here)"
                                           if args.scaling == "strong" else "every rank scores its own all-vs-all job"),
                              **collective_note(args, world, use_dist)})
        result["roofline"] = roofline_block("c3", {"env": "k_env_group", "sweep": "k_sweep_duo<8, 16 lanes, 240 events> (four pairs per wavefront)"}[dom], algo, phase_ms[dom], step_ms=result["ms_per_step"])
        result["kernel_ms"] = phase_ms
        if emulated:
            emulated["single_gpu_ms"] = elapsed / args.steps * 1e3
            emulated["estimated_speedup_without_collective"] = emulated["single_gpu_ms"] / max(emulated["per_rank_ms"])
            emulated["estimated_speedup_median_rank"] = emulated["single_gpu_ms"] / float(np.median(emulated["per_rank_ms"]))
            emulated["note"] = "the W ranks' shares timed one after the other on ONE GPU, no collective: an estimate, not a measurement on W GPUs"
            result["extras"] = {"emulated_strong_scaling": emulated}
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as orc  # checker / baseline only

            cores = usable_cores()
            lo = orc.LoCoHD(w["types"], orc.WeightFunction("uniform", [3.0, 10.0]), orc.TagPairingRule({"accept_same": False}), n_of_threads=cores)
            scores = out.cpu().numpy()
            local = np.stack([la, la], 1).astype(np.int64)
            t1, err, done = time.perf_counter(), 0.0, 0
            for k, (a, b) in enumerate(w["spairs"]):
                if k % 41:
                    continue  # a spread of 30 decoy pairs
                got = np.asarray(lo.from_arrays(w["decoys"][a][0], w["decoys"][a][1], w["decoys"][a][2], w["decoys"][b][0], w["decoys"][b][1],
                                                w["decoys"][b][2], local, w["thr"]))
                err = max(err, float(np.max(np.abs(got - scores[k * len(la):(k + 1) * len(la)]))))
                done += len(la)
            dt = time.perf_counter() - t1
            result["cpu_baseline"] = {"value": done / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
                                      "sample": f"every 41st decoy pair ({done} anchor pairs), {dt:.1f} s, C oracle with {cores} pthreads, one "
                                                f"from_primitives call per decoy pair like the reference's loop"}
            result["max_abs_err_vs_cpu"] = err
            result["parity_sample_pairs"] = done
            if not (err <= 1e-6):
                result["parity_failed"] = True
    for s_ in sessions:
        s_.close()
    return result


# ---------------------------------------------------------------------------------------------------------------------------
# C2b: dense from_coords
# ---------------------------------------------------------------------------------------------------------------------------
def run_c2b(args, torch, dist, dev, rank, world, use_dist):
    import loco_hd_amd as lh
    from loco_hd_amd.device import DeviceSession

    w = make_workload("c2b", rank, 0, n_atoms=args.dense_atoms)
    lchd = lh.LoCoHD([f"c{i}" for i in range(w["C"])], lh.WeightFunction(*w["wf"]))
    sess = DeviceSession(lchd, device=dev.index)
    sess.enable_timing(True)
    a, b = sess.upload(w["xyz_a"], w["cat_a"]), sess.upload(w["xyz_b"], w["cat_b"])
    n = w["n"]
    out = torch.empty(n, dtype=torch.float64, device=dev)
    phase_ms = {"env": 0.0, "sweep": 0.0}
    collect = [False]

    def step():
        sess.from_coords(a, b, out=out)
        if collect[0]:
            for k in phase_ms:
                phase_ms[k] += sess.last_ms()[k]

    def start_collect():
        collect[0] = True

    elapsed = Harness(args, torch, dist, dev, use_dist).run(step, on_timed_start=start_collect)
    phase_ms = {k: v / max(args.steps, 1) for k, v in phase_ms.items()}
    result = None
    if rank == 0:
        algo = n * (2 * n * 28 + 16)  # every pair reads both whole structures: B_pair = (n_A + n_B) * 28 + 16
        dom = max(phase_ms, key=lambda k: phase_ms[k])
        result = base_result(args, world, n * world, elapsed, w["label"], {"pairs_per_gpu": n, "mean_env_points_per_pair": 2 * n,
                                                                           "sharding": "replicas only (one from_coords call per rank)"})
        result["scaling"] = "weak"
        fused = sess.last_dense_fused()
        kname = "k_dense_fused (row sort + sweep of a row pair in one workgroup)" if fused else {"env": "k_env_rows2 (both structures' rows in one launch)", "sweep": "k_sweep"}[dom]
        result["roofline"] = roofline_block("c2b" if n == 10_000 else f"c2b_{n}", kname, algo, phase_ms[dom], step_ms=result["ms_per_step"])
        result["config"]["dense_path"] = "fused (one launch, scores only leave the CU)" if fused else "row sort to the environment store, then sweep"
        result["kernel_ms"] = phase_ms
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as orc  # checker / baseline only

            cores = usable_cores()
            scores = out.cpu().numpy()
            names = [f"c{i}" for i in range(w["C"])]
            sa, sb = [names[k] for k in w["cat_a"]], [names[k] for k in w["cat_b"]]
            # the reference's from_coords (src/locohd.rs:463-476): both distance matrices first, serially (utils.rs:10-22), then the
            # rows -- stable sort with the labels + sweep -- over its thread pool (:434-446).  Sample: rows spread over the structure.
            n_rows = min(n, max(48, 16 * cores))
            rows = np.unique(np.linspace(0, n - 1, n_rows).astype(np.int64))

            def dist_rows(x):  # utils.rs:1-8 order
                d = x[rows][:, None, :] - x[None, :, :]
                return np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])

            t0 = time.perf_counter()
            da, db = dist_rows(w["xyz_a"]), dist_rows(w["xyz_b"])
            t_dist = time.perf_counter() - t0
            res = {}
            for threads in (cores, 1):
                lo = orc.LoCoHD(names, orc.WeightFunction(*w["wf"]), n_of_threads=threads)
                sub = slice(None) if threads > 1 else slice(0, max(8, len(rows) // cores))
                t1 = time.perf_counter()
                got_cpu = np.asarray(lo.from_dmxs(sa, sb, da[sub], db[sub]))
                res[threads] = (len(got_cpu) / (time.perf_counter() - t1), got_cpu)
            err = float(np.max(np.abs(res[cores][1] - scores[rows])))
            rate_all = len(rows) / (len(rows) / res[cores][0] + t_dist)
            result["cpu_baseline"] = {"value": rate_all, "unit": "pairs/s", "cores": cores, "kind": "port",
                                      "value_1thread": res[1][0],
                                      "sample": f"{len(rows)} rows spread over the structure: their {2 * len(rows)} distance rows by NumPy on one "
                                                f"core ({t_dist:.2f} s: the reference builds its matrices serially, utils.rs:10-22), then the C oracle's "
                                                f"stable row sort + sweep over {cores} threads ({len(rows) / res[cores][0]:.2f} s; one thread: "
                                                f"{res[1][0]:.1f} rows/s)"}
            result["max_abs_err_vs_cpu"] = float(err)
            result["parity_sample_pairs"] = len(rows)
            if not (err <= 1e-6):
                result["parity_failed"] = True
    sess.close()
    return result


# ---------------------------------------------------------------------------------------------------------------------------
# C2a / C5: pair lists
# ---------------------------------------------------------------------------------------------------------------------------
def run_pairs(args, torch, dist, dev, rank, world, use_dist):
    import loco_hd_amd as lh
    from loco_hd_amd.device import DeviceSession
    from loco_hd_amd.dist import gather_scores, select_shard, unshard

    strong = args.scaling == "strong"
    w = make_workload(args.workload, rank, args.pairs, same_on_all_ranks=strong)
    lchd = lh.LoCoHD([f"c{i}" for i in range(w["C"])], lh.WeightFunction(*w["wf"]))
    # Two sessions (contexts with their own workspace) take the steps in turn: step k+1 is enqueued while step k still runs, so
    # the GPU does not idle during the status hand-over and the host's launch work of a step (a step is still one complete
    # pass; both sessions enqueue on the same stream, so the passes themselves run one after the other).
    # --streams 2 (default for N > 1): each session enqueues on a stream of its own, so the kernels of step k+1 fill the launch
    # gaps and the tail of step k (independent passes; every pass and every collective of the timed steps still completes inside
    # the timed region).  --streams 1 (default at N = 1): both sessions share one stream, the passes run strictly one after the
    # other and the per-kernel event times are those of the kernels alone (the roofline figures need that).
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)] if args.streams == 2 else [torch.cuda.current_stream(dev)] * 2
    sessions = []
    for st_ in streams:
        with torch.cuda.stream(st_):
            sessions.append(DeviceSession(lchd, device=dev.index))  # (binds the session to the current stream)
    sess = sessions[0]
    clouds = [(s_.upload(w["xyz_a"], w["cat_a"]), s_.upload(w["xyz_b"], w["cat_b"])) for s_ in sessions]
    torch.cuda.synchronize()
    cloud_a, cloud_b = clouds[0]
    anchors = torch.from_numpy(w["pairs"]).to(dev)
    p = anchors.shape[0]
    for s_ in sessions:
        s_.enable_timing(True)
    state = {}

    def run_rank(vrank, vworld, tag, collective):
        """`collective`: the real gather (use_dist) -- an emulated rank only copies into its gather slot"""
        counter = [0]
        phase_ms = {"cells": 0.0, "anchors": 0.0, "env": 0.0, "sweep": 0.0, "shard": 0.0}
        collect = [False]
        in_flight, pending = [False, False], [None, None]
        if strong:
            slot = int(p / vworld * 1.5) + 4096  # gather slot per rank (the shares differ by a bin's worth of pairs)
            locals_ = [torch.zeros((2, slot), dtype=torch.float64, device=dev) for _ in range(2)]
            sels, planned = [None, None], [None, None]
            gathered = [torch.empty(vworld * 2 * slot, dtype=torch.float64, device=dev) if (collective and rank == 0) else None for _ in range(2)]
            full = torch.empty(p, dtype=torch.float64, device=dev)
            counts_seen = [None, None]
        else:
            outs = [torch.empty(p, dtype=torch.float64, device=dev) for _ in range(2)]
            gathered = [torch.empty(p * vworld, dtype=torch.float64, device=dev) if (collective and rank == 0) else None for _ in range(2)]
        torch.cuda.synchronize()  # (the buffers above were filled on the default stream; the steps run on the sessions' streams)

        def finish(k):  # wait for the pass enqueued on session k (raises on a device-side error), book its kernel times
            if in_flight[k]:
                sessions[k].finish()
                in_flight[k] = False
                if collect[0]:
                    for name, v in sessions[k].last_ms().items():
                        phase_ms[name] += v

        def wait_gather(k):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
                if strong and rank == 0 and collective:  # restore anchor-pair order (inside the step that owns the gather)
                    unshard(gathered[k], counts_seen[k], slot, p, out=full, session=sessions[k])

        def step():
            k = counter[0] % 2
            counter[0] += 1
            with torch.cuda.stream(streams[k]):
                step_on(k)

        def step_on(k):
            finish(k)  # the session's previous pass (two steps ago) and its buffers
            wait_gather(k)
            if strong:
                t0 = time.perf_counter()
                # the partition of a pair list is a pure function of the list: planned once per (list, rank, session), reused by
                # every later step on the unchanged tensor (dist.select_shard(cache=True)) -- no partition kernels in the steady state
                sel, idx, counts = select_shard(anchors, w["n"], vworld, vrank, session=sessions[k], n_atoms_b=w["n"], cache=not args.no_plan_cache)
                n_mine = counts[vrank]
                assert n_mine <= slot, (n_mine, slot)
                if planned[k] is not sel:  # first step of this session (or the cache is off): positions into the gather buffer
                    locals_[k][1, :n_mine] = idx.view(torch.float64)
                    planned[k] = sel
                sels[k] = sel
                if collect[0]:
                    phase_ms["shard"] += (time.perf_counter() - t0) * 1e3
                counts_seen[k] = counts
                sessions[k].from_primitives_async(clouds[k][0], clouds[k][1], sels[k], w["thr"], locals_[k][0])
                in_flight[k] = True
                if collective and args.gather == "step":
                    pending[k] = gather_scores(locals_[k], gathered[k], vworld, rank, force_collective=True, async_op=True)
                state[tag + "_n_mine"] = n_mine
            else:
                sessions[k].from_primitives_async(clouds[k][0], clouds[k][1], anchors, w["thr"], outs[k])
                in_flight[k] = True
                if collective and args.gather == "step":
                    pending[k] = gather_scores(outs[k], gathered[k], vworld, rank, force_collective=True, async_op=True)

        def drain():
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    finish(k)
                    wait_gather(k)

        def start_collect():
            collect[0] = True

        elapsed = Harness(args, torch, dist, dev, use_dist).run(step, drain, start_collect)
        collect[0] = False
        kl = (counter[0] - 1) % 2
        state[tag] = dict(phase_ms={k: v / max(args.steps, 1) for k, v in phase_ms.items()}, kl=kl,
                          out=(locals_[kl] if strong else outs[kl]), gathered=gathered[kl], full=(full if strong else None),
                          counts=(counts_seen[kl] if strong else None), slot=(slot if strong else p))
        return elapsed

    emulated = None
    if strong and args.emulate_world > 1 and world == 1:
        per_rank, per_phase, per_n = [], [], []
        for r in range(args.emulate_world):
            per_rank.append(run_rank(r, args.emulate_world, f"r{r}", collective=False) / args.steps * 1e3)
            per_phase.append(state[f"r{r}"]["phase_ms"])
            per_n.append(state[f"r{r}_n_mine"])
        emulated = {"world": args.emulate_world, "per_rank_ms": per_rank, "per_rank_pairs": per_n, "per_rank_kernel_ms": per_phase}
    elapsed = run_rank(rank, world, "main", collective=use_dist)
    st = state["main"]
    phase_ms = st["phase_ms"]
    total_pairs = p if strong else p * world

    # Outside the timed region: the same clouds with every anchor used ONCE (pairs (i, i), i < N) -- no environment is
    # shared between pairs, so this is the per-call cost of a plain from_primitives(i, i) call with device-resident inputs.
    extras = {}
    if emulated:
        emulated["single_gpu_ms"] = elapsed / args.steps * 1e3
        emulated["estimated_speedup_without_collective"] = emulated["single_gpu_ms"] / max(emulated["per_rank_ms"])
        emulated["estimated_speedup_median_rank"] = emulated["single_gpu_ms"] / float(np.median(emulated["per_rank_ms"]))
        emulated["note"] = "the W ranks' shares timed one after the other on ONE GPU, no collective: an estimate, not a measurement on W GPUs"
        extras["emulated_strong_scaling"] = emulated
    if rank == 0 and args.workload == "c2a" and not args.no_cpu_baseline and not strong:  # profiling runs launch only the timed steps
        n_atoms = w["n"]
        uniq = anchors[:n_atoms].contiguous()
        out_u = torch.empty(n_atoms, dtype=torch.float64, device=dev)
        for _ in range(3):
            sess.from_primitives(cloud_a, cloud_b, uniq, w["thr"], out=out_u)
        torch.cuda.synchronize()
        tu = time.perf_counter()
        for _ in range(20):
            sess.from_primitives(cloud_a, cloud_b, uniq, w["thr"], out=out_u)
        torch.cuda.synchronize()
        tu = (time.perf_counter() - tu) / 20
        extras["unique_anchor_call"] = {"pairs": int(n_atoms), "ms_per_call": tu * 1e3, "pairs_per_s": n_atoms / tu,
                                        "algorithmic_bytes": int(sess.last_env_points()) * 28 + 16 * int(n_atoms),
                                        "note": "every anchor used once: no environment re-use between pairs (whole call: prologue, environments, "
                                                "sweep, status wait)"}
    if rank == 0 and world == 1 and args.streams == 1 and not args.no_cpu_baseline and not strong and not emulated:
        # Outside the timed region as well: the same steps with the two alternating sessions on a stream EACH -- consecutive steps then
        # overlap (the next step's cell lists and environments under the tail of this step's sweep).  Reported next to the headline, not as
        # it: roofline.achieved / frac are defined per launch of the dominant kernel, and a launch that shares the chip has no time of its own.
        try:
            st2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
            ss2 = []
            for q_ in st2:
                with torch.cuda.stream(q_):
                    ss2.append(DeviceSession(lchd, device=dev.index))
            cl2 = [(s_.upload(w["xyz_a"], w["cat_a"]), s_.upload(w["xyz_b"], w["cat_b"])) for s_ in ss2]
            ou2 = [torch.empty(p, dtype=torch.float64, device=dev) for _ in range(2)]
            torch.cuda.synchronize()
            fly = [False, False]

            def overlapped(n_steps):
                for k_ in range(n_steps):
                    j_ = k_ % 2
                    with torch.cuda.stream(st2[j_]):
                        if fly[j_]:
                            ss2[j_].finish()
                        ss2[j_].from_primitives_async(cl2[j_][0], cl2[j_][1], anchors, w["thr"], ou2[j_])
                        fly[j_] = True
                for j_ in range(2):
                    with torch.cuda.stream(st2[j_]):
                        if fly[j_]:
                            ss2[j_].finish()
                            fly[j_] = False
                torch.cuda.synchronize()

            overlapped(PRIME_STEPS + 2)
            t2 = time.perf_counter()
            overlapped(args.steps)
            t2 = (time.perf_counter() - t2) / max(args.steps, 1)
            same2 = bool(torch.equal(ou2[0], ou2[1]))
            extras["two_streams"] = {"ms_per_step": t2 * 1e3, "pairs_per_s": p / t2, "steps": args.steps, "both_sessions_same_scores": same2,
                                     "note": "the timed steps again, the two alternating sessions on a stream each: complete, independent passes that "
                                             "overlap; NOT the headline (see config.streams_note)"}
            for s_ in ss2:
                s_.close()
        except Exception as exc:  # (an extra: never the reason for a missing bench line)
            extras["two_streams"] = {"error": repr(exc)}
    value_incl = None
    if rank == 0 and not args.no_cpu_baseline and not strong:
        # the same job through the host-pointer entry point (lchd_from_primitives: packing into the pinned
        # staging block + ONE H2D copy of structures and pairs + the pass + ONE D2H copy of the scores, all inside the call)
        pa = lh.api._Packed(w["xyz_a"], w["cat_a"], np.zeros(w["n"], np.int32))
        pb = lh.api._Packed(w["xyz_b"], w["cat_b"], np.zeros(w["n"], np.int32))
        for _ in range(2):  # (first call: the context's staging block grows; second: the launch set picked from the first pass's pair statistics)
            host_scores = lchd.from_packed(pa, pb, w["pairs"], w["thr"])
        ths = []
        for _ in range(5):
            th = time.perf_counter()
            host_scores = lchd.from_packed(pa, pb, w["pairs"], w["thr"])
            ths.append(time.perf_counter() - th)
        th = float(np.median(ths))
        value_incl = {"value": p / th, "unit": "pairs/s", "ms_per_call": th * 1e3,
                      "what": "LoCoHD.from_packed on host arrays (median of 5 calls): staging + H2D of both structures and the pair list + pass + D2H of the "
                              "scores; above 2^18 pairs the list and the scores travel in pipelined chunks copied by up to four threads"}
        extras["host_call_scores_equal_device_resident"] = None  # filled below
    # environment points of this rank's pairs (algorithmic bytes), and the scores to check
    if strong:
        n_mine = state["main_n_mine"]
        sel, idx, _ = select_shard(anchors, w["n"], world, rank, session=sess, n_atoms_b=w["n"])
        check = torch.empty(n_mine, dtype=torch.float64, device=dev)
        sess.from_primitives(cloud_a, cloud_b, sel, w["thr"], out=check)
        env_points = sess.last_env_points()
        pairs_this_rank = n_mine
        if world == 1:
            scores = torch.empty(p, dtype=torch.float64, device=dev)
            scores[idx] = check
            scores = scores.cpu().numpy()
        elif rank == 0 and st["gathered"] is not None and args.gather == "step":
            scores = st["full"].cpu().numpy()
        else:
            scores = None
    else:
        sess.from_primitives(cloud_a, cloud_b, anchors, w["thr"], out=st["out"])
        env_points = sess.last_env_points()
        pairs_this_rank = p
        scores = st["out"].cpu().numpy()

    final_line = None
    if rank == 0:
        algo_bytes = env_points * 28 + 16 * pairs_this_rank  # B_pair = (n_A + n_B) * 28 B + 8 B + 8 B
        dom = max(("env", "sweep"), key=lambda k: phase_ms[k])
        dom_name = {"env": "k_env_group (side A + side B in one launch)",
                    "sweep": "k_sweep_duo<slots, 32 lanes, 480 events> (two pairs per wavefront; chunk-start counts from the environments' prefix-count rows up to 16 slots; its launch, the k_pair_meta record pass in front of it and the INDIRECT sweep for the few larger pairs)"}[dom]
        result = base_result(args, world, total_pairs, elapsed, w["label"],
                             {"pairs_total" if strong else "pairs_per_gpu": p, "pairs_this_rank": pairs_this_rank,
                              "mean_env_points_per_pair": env_points / max(pairs_this_rank, 1),
                              "sharding": ("pairs binned by their side-A anchor (library kernels, every rank the same rule), every rank holds both structures"
                                           if strong else "every rank scores its own pair list of the same two structures"),
                              **collective_note(args, world, use_dist)})
        result["roofline"] = roofline_block(args.workload, dom_name, algo_bytes, phase_ms[dom], step_ms=result["ms_per_step"])
        result["kernel_ms"] = phase_ms
        # how often the step uses an environment it builds: frac (28 B per environment point PER PAIR) can exceed 1 when this is large
        pl = w["pairs"]
        result["config"]["env_reuse_factor"] = 2.0 * len(pl) / max(len(np.unique(pl[:, 0])) + len(np.unique(pl[:, 1])), 1)
        result["config"]["env_reuse_note"] = ("anchor pairs per distinct anchor of the list (SURVEY.md 8d defines C2a as 100 permutation rounds over all "
                                              "atoms): distances + sort run once per distinct anchor, the sweep once per pair; roofline.frac_no_reuse is "
                                              "the same byte model on the list with factor 1")
        if "unique_anchor_call" in extras:
            u = extras["unique_anchor_call"]
            result["roofline"]["frac_no_reuse"] = u["algorithmic_bytes"] / (u["ms_per_call"] * 1e-3) / (HBM_PEAK_GBS * 1e9)
        if strong:
            # what the plan cache keeps OUT of the timed steps: one uncached plan + select of this rank's share (two partition kernels and
            # the host's wait for the counts), timed once here so that cached and uncached (--no-plan-cache) lines can be compared
            torch.cuda.synchronize()
            t_plan = time.perf_counter()
            select_shard(anchors, w["n"], world, rank, session=sess, n_atoms_b=w["n"], cache=False)
            torch.cuda.synchronize()
            result["config"]["plan_cache"] = not args.no_plan_cache
            result["config"]["plan_ms"] = (time.perf_counter() - t_plan) * 1e3
            result["config"]["plan_note"] = ("the partition of the unchanged pair list is planned once and reused (dist.select_shard(cache=True)): "
                                             "plan_ms is NOT inside ms_per_step" if not args.no_plan_cache else "planned in every timed step")
        result["extras"] = extras
        if value_incl is not None:
            result["value_incl_h2d_d2h"] = value_incl
            extras["host_call_scores_equal_device_resident"] = bool(scores is not None and np.array_equal(host_scores, scores))
        if not args.no_cpu_baseline and world == 1 and scores is not None:  # the CPU leg (and the parity gate on its sample) runs at N = 1 only
            base, err, m = cpu_baseline(w, scores)
            result["cpu_baseline"] = base
            result["max_abs_err_vs_cpu"] = err
            result["parity_sample_pairs"] = m
            if not (err <= 1e-6):
                result["parity_failed"] = True
        final_line = result
    if use_dist:
        if strong:
            if rank == 0 and scores is not None:  # the restored vector against rank 0's own scoring of the whole list
                ref = torch.empty(p, dtype=torch.float64, device=dev)
                sess.from_primitives(cloud_a, cloud_b, anchors, w["thr"], out=ref)
                assert torch.equal(torch.from_numpy(scores).to(dev), ref), "gathered + restored scores differ from a single-GPU pass"
        else:
            kl = st["kl"]
            if args.gather == "end":  # one gather of the last step's slices, outside the timed region
                h = gather_scores(st["out"], st["gathered"], world, rank, force_collective=True, async_op=True)
                if h is not None:
                    h.wait()
                torch.cuda.synchronize()
            if rank == 0 and st["gathered"] is not None:
                assert torch.equal(st["gathered"][:p], st["out"]), "gathered scores differ from the local ones"
        dist.barrier()
    for s_ in sessions:
        s_.close()
    return final_line


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start `python -m torch.distributed.run --nproc-per-node N bench.py
    <the same arguments>` as a child process -- this process has not imported torch and has made no GPU call, and it never
    replaces itself --, pass the child's stdout through line by line (rank 0's JSON line stays the last line) and return its
    exit code.  The rendezvous is on 127.0.0.1 at a port that was free a moment ago (MASTER_PORT overrides)."""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between the ranks needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for line in child.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
        return child.wait()
    except BaseException:
        child.terminate()  # (the exact process this function started; torchrun takes its workers down with it)
        try:
            child.wait(timeout=30)
        except subprocess.TimeoutExpired:
            child.kill()
        raise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2a", choices=["c2a", "c2b", "c3", "c4", "c5"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="multi-GPU: strong (default for --gpus N > 1: BASELINE.json's 8-GPU configurations are fixed-size jobs and its target "
                         "is '>= 6x at 8 GPUs') = the named workload in total, sharded across the GPUs; weak = the named workload per GPU")
    ap.add_argument("--frames", type=int, default=5000, help="c4: frames of the trajectory")
    ap.add_argument("--chunk", type=int, default=2500, help="c4: frames per scoring pass (measured, ms per step of 5000 frames: 1250 6.26, 2500 5.88, 5000 5.81 -- one pass, nothing streamed)")
    ap.add_argument("--c4-sessions", type=int, default=2, choices=[1, 2],
                    help="c4: sessions the chunks of a step alternate between (2: asynchronous passes, the next chunk converted on a side stream)")
    ap.add_argument("--pairs", type=int, default=1_000_000, help="c2a / c5: anchor pairs (per GPU when weak, in total when strong)")
    ap.add_argument("--gather", default=None, choices=["end", "step"],
                    help="multi-GPU: RCCL gather of the scores to rank 0 inside every timed step (default for N > 1) or once behind the timed region")
    ap.add_argument("--no-plan-cache", action="store_true", help="strong scaling: plan and select the rank's share in every step (two partition kernels)")
    ap.add_argument("--emulate-world", type=int, default=0, help="one GPU, --scaling strong: time every rank's share of a W-GPU job in turn")
    ap.add_argument("--dense-atoms", type=int, default=10_000, help="c2b: atoms per structure (= anchor pairs; the environment is the whole structure)")
    ap.add_argument("--streams", type=int, default=None, choices=[1, 2],
                    help="c2a / c5 / c3: HIP streams the two alternating sessions enqueue on (default: 1 at N = 1 -- clean per-kernel times --, "
                         "2 for N > 1 and --emulate-world: consecutive steps overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg, the parity gate and the extras (profiling runs)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (a CHILD process, before this one has
        # imported torch or touched a GPU -- never an exec), relay their output and exit with the launcher's code
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        print(f"[bench] rank {rank} of {world} up (pid {os.getpid()}, local rank {local_rank})", file=sys.stderr, flush=True)
    if world != args.gpus:  # (the launcher's world is what runs; the defaults below follow it, not just --gpus)
        if args.gpus != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch as many ranks as GPUs")
        args.gpus = world  # torchrun without --gpus: the job is `world` GPUs wide
    if args.gather is None:
        args.gather = "step"
    if args.streams is None:
        args.streams = 2 if (args.gpus > 1 or args.emulate_world > 1) else 1
    if args.scaling is None:  # c2b has no sharding (one from_coords call): replicas only
        args.scaling = "strong" if (args.gpus > 1 or args.emulate_world > 1) and args.workload != "c2b" else "weak"

    import torch
    import torch.distributed as dist

    if os.environ.get("LCHD_BENCH_RENDEZVOUS_ONLY"):
        # launcher check (tests/test_bench_helpers.py, no GPU): the ranks meet over gloo, add up their ranks, rank 0 reports
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "world": world, "rank_sum": float(t.item()), "gpus": args.gpus}), flush=True)
        return
    n_dev = torch.cuda.device_count()  # (does not initialise the GPU on this image)
    # LCHD_BENCH_SHARE_GPU=1: a REHEARSAL of the N-rank job on fewer GPUs (tests/test_gpu_dist.py: two ranks on the one GPU of a
    # test box) -- ranks share devices round-robin and meet over gloo with a host-staged gather (RCCL refuses two ranks on one
    # device); every code path of a real N-rank run except the RCCL transport, and its line says so: not a scaling measurement
    share = bool(os.environ.get("LCHD_BENCH_SHARE_GPU")) and world > 1 and n_dev >= 1
    if n_dev <= local_rank and not share:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but this machine shows {n_dev} GPU(s): "
                         f"--gpus {args.gpus} needs {args.gpus} visible devices (one rank per GPU)")
    dev_index = local_rank % n_dev if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or bool(os.environ.get("LCHD_BENCH_FORCE_DIST"))  # the latter: exercise RCCL init + gather on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    runner = {"c4": run_c4, "c3": run_c3, "c2b": run_c2b}.get(args.workload, run_pairs)
    result = runner(args, torch, dist, dev, rank, world, use_dist)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and share:
        result["rehearsal"] = (f"{world} ranks shared {n_dev} GPU(s) over gloo with a host-staged gather (LCHD_BENCH_SHARE_GPU): every code path "
                               "of an N-rank run except the RCCL transport; the value is NOT a scaling measurement")
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so that the JSON line is the LAST line of stdout
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(result), flush=True)
        if result.get("parity_failed"):
            sys.exit(3)


if __name__ == "__main__":
    main()
