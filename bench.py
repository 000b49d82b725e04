#!/usr/bin/env python3
"""bench.py -- anchor-pair LoCoHD scores/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (SURVEY.md section 8d, config C2a = BASELINE.json configs[1]): two 10 000-atom random clouds at
protein-like density 0.05 atoms/A^3 (box side 58.5 A), 10 primitive categories, hyper_exp[1.0, 0.1] weight
function, Hellinger-2, from_primitives semantics with threshold 10 A, P = 10^6 anchor pairs per GPU:
pair k = (k mod N, perm_r(k mod N)) with r = k div N and perm_r a seeded permutation (round 0 = identity), so
every one of the 10^4 atoms of each cloud is an anchor ~100 times.  Coordinates, categories and anchor pairs
are resident in HBM before the timed region; ONE STEP = one full pass of the hot path over that batch:
cell lists for both clouds, anchor de-duplication, environment build (radius search + f64 distances + sort),
merge sweep (Hellinger + CDF + reduction), status read-back.  Nothing is cached across steps.  Two sessions take the steps
in turn so that step k+1 is already enqueued while step k runs (the GPU does not idle during the host's status read-back).

Multi-GPU (weak scaling): every rank holds both clouds and its own 10^6 pairs (different permutation
rounds) and keeps its scores in its own HBM: anchor pairs are independent, the path has no exchange step, so the timed
steps contain no collective.  After the timed region the score slices are gathered to rank 0 ONCE over RCCL and checked
(`--gather step` puts an asynchronous gather into every timed step instead: the per-call delivery of all scores to one rank).

The JSON line also carries:
  roofline      dominant kernel's ALGORITHMIC bytes per launch (B_pair = (n_A+n_B)*28 + 16)
                / its average launch duration (HIP events on the launch stream, recorded by the C library)
                against the 8 TB/s HBM peak.
  cpu_baseline  the CPU oracle (C restatement of the reference algorithm, kind "port": the Rust reference
                cannot be built here) timed on this host's cores on a bounded sample of the same pairs;
                the same sample is the parity gate (max |gpu - cpu| must be <= 1e-6).
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


def make_workload(name: str, rank: int, n_pairs: int):
    """Synthetic inputs of SURVEY.md 8(d).  Returns dict(xyz_a, xyz_b, cat_a, cat_b, pairs, thr, C, wf, label)."""
    if name == "c2a":
        n, c, seed, thr = 10_000, 10, 2, 10.0
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
    prng = np.random.default_rng(1000 * seed + rank)  # every rank scores different pairs of the same clouds
    if rounds:
        idx = np.arange(n, dtype=np.int64)
        cols = []
        for r in range(rounds):
            perm = idx if (r == 0 and rank == 0) else prng.permutation(n)
            cols.append(np.stack([idx, perm], 1))
        pairs = np.concatenate(cols, 0)[:n_pairs]
        label = (f"C2a: 2x{n}-atom random clouds (0.05 atoms/A^3), {c} categories, hyper_exp[1,0.1], Hellinger-2, "
                 f"from_primitives thr {thr:g} A, {len(pairs)} anchor pairs/GPU = {rounds} permutation rounds over all atoms")
    else:
        pairs = np.stack([prng.integers(0, n, n_pairs), prng.integers(0, n, n_pairs)], 1).astype(np.int64)
        label = f"C5: 2x{n}-point clouds, {c} categories, hyper_exp[1,0.1], {n_pairs} random anchor pairs/GPU, thr {thr:g} A"
    return dict(xyz_a=xyz_a, xyz_b=xyz_b, cat_a=cat_a, cat_b=cat_b, pairs=np.ascontiguousarray(pairs), thr=thr, C=c,
                wf=("hyper_exp", [1.0, 0.1]), label=label, n=n)


def measured_traffic(kernel_name: str, workload_label: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/), if they were
    taken for this kernel and workload; PMC counters cannot be read from inside this process."""
    try:
        t = json.loads((ROOT / "profiles" / "r01" / "traffic_c2a.json").read_text())
    except (OSError, ValueError):
        return None, None
    if t.get("kernel") in kernel_name and workload_label.startswith(t.get("workload_prefix", "\0")):
        return t.get("traffic_bytes_per_launch"), "profiles/r01/traffic_c2a.json (separate rocprofv3 --pmc passes of this command)"
    return None, None


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
    return {"value": m / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"first {m} anchor pairs of the same workload, {dt:.1f} s, C oracle (oracle/locohd_oracle.c) with "
                      f"{cores} pthreads over anchor pairs; reference Rust core not buildable in this image"}, err, m


def run_c4(args, torch, dist, dev, rank, world, use_dist):
    """BASELINE.json configs[3] (MD-trajectory mode) with the structure -> primitive-atom step on the device as well:
    one reference structure vs F frames of the same system, per-residue "Cent" anchors, accept_same=False, uniform[3,10],
    threshold 10 A (python_codes/trajectory_analyzer.py:76-119).  The frames are float32 SOURCE atoms resident in HBM
    ([F][5336][3]); one step = for every chunk of frames: k_frames_centroids (2001 primitive atoms per frame from their
    source atoms, np.mean arithmetic) + the full scoring pass (cell lists, environments, sweep) of the chunk's anchor pairs."""
    import loco_hd_amd as lh
    from loco_hd_amd.device import DeviceSession

    n_res, per_res, n_frames, chunk = 667, 8, args.frames, args.chunk
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
    gen = torch.Generator(device=dev).manual_seed(4 + rank)
    d_ref = torch.from_numpy(ref_atoms).to(dev)
    frames = (d_ref[None] + 0.5 * torch.randn((n_frames, n_src, 3), generator=gen, device=dev, dtype=torch.float32)).contiguous()
    chunk = min(chunk, n_frames)
    buf = sess.frames_buffer(ref, chunk)
    sess.set_frame_sources(buf, topo)
    la = torch.arange(0, n_prim, 3, dtype=torch.int64, device=dev)
    offs = torch.arange(chunk, dtype=torch.int64, device=dev).repeat_interleave(len(la)) * n_prim
    anchors = torch.stack([la.repeat(chunk), la.repeat(chunk) + offs], 1).contiguous()
    p = n_frames * len(la)
    out = torch.empty(p, dtype=torch.float64, device=dev)
    starts = list(range(0, n_frames, chunk))
    phase = {"convert": 0.0, "cells": 0.0, "anchors": 0.0, "env": 0.0, "sweep": 0.0}
    env_points = [0]

    def step(collect=False):
        for f0 in starts:
            nf = min(chunk, n_frames - f0)
            sess.load_atom_frames_dev(buf, frames[f0:f0 + nf])
            sess.from_primitives(ref, buf, anchors[: nf * len(la)], 10.0, out=out[f0 * len(la):(f0 + nf) * len(la)])
            if collect:
                phase["convert"] += sess.last_convert_ms(buf)
                for k, v in sess.last_ms().items():
                    phase[k] += v

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(collect=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
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
        achieved = algo / (phase[dom] * 1e-3) / 1e9
        conv_bytes = n_frames * (n_src * 12 + n_prim * 24)  # every source atom read once (3 x f32), every primitive atom written (3 x f64)
        label = (f"C4: 1 reference vs {n_frames} frames of a {n_prim}-primitive-atom system ({n_src} float32 source atoms/frame, "
                 f"converted on the device), {len(la)} per-residue anchors/frame, accept_same=False, uniform[3,10], thr 10 A, "
                 f"chunks of {chunk} frames")
        result = {
            "metric": "anchor-pair LoCoHD scores/sec", "value": p * world * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "pairs_per_gpu": p, "mean_env_points_per_pair": env_points[0] / p},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": {"env": "k_env_cells (side A + side B)", "sweep": "k_sweep"}[dom],
                         "algorithmic_bytes_per_launch": algo / len(starts), "avg_launch_ms": phase[dom] / len(starts)},
            "kernel_ms": phase,
            "extras": {"k_frames_centroids": {"ms_per_step": phase["convert"], "algorithmic_bytes_per_step": conv_bytes,
                                              "GB_per_s": conv_bytes / max(phase["convert"], 1e-9) / 1e6,
                                              "frac_of_hbm_peak": conv_bytes / max(phase["convert"], 1e-9) / 1e6 / HBM_PEAK_GBS}},
        }
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
    sess.close()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2a", choices=["c2a", "c5", "c4"])
    ap.add_argument("--frames", type=int, default=5000, help="c4: frames of the trajectory")
    ap.add_argument("--chunk", type=int, default=1250, help="c4: frames per scoring pass")
    ap.add_argument("--pairs", type=int, default=1_000_000, help="anchor pairs per GPU per step")
    ap.add_argument("--gather", default="end", choices=["end", "step"],
                    help="multi-GPU: gather the score slices to rank 0 once after the timed region (default: the path itself needs no "
                         "collective) or asynchronously inside every timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg, the parity gate and the extras (profiling runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or bool(os.environ.get("LCHD_BENCH_FORCE_DIST"))  # the latter: exercise RCCL init + gather on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload == "c4":  # trajectory mode: frames shard across ranks with no exchange step (weak scaling: F frames per rank)
        result = run_c4(args, torch, dist, dev, rank, world, use_dist)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            import ctypes

            try:
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            print(json.dumps(result), flush=True)
            if result.get("parity_failed"):
                sys.exit(3)
        return

    import loco_hd_amd as lh
    from loco_hd_amd.device import DeviceSession
    from loco_hd_amd.dist import gather_scores

    w = make_workload(args.workload, rank, args.pairs)
    lchd = lh.LoCoHD([f"c{i}" for i in range(w["C"])], lh.WeightFunction(*w["wf"]))
    # Two sessions (contexts with their own workspace) take the steps in turn: step k+1 is enqueued while step k still runs, so
    # the GPU does not idle during the status read-back and the host's launch work of a step (a step is still one complete
    # pass; both sessions enqueue on the same stream, so the passes themselves run one after the other).
    sessions = [DeviceSession(lchd, device=local_rank) for _ in range(2)]
    sess = sessions[0]
    clouds = [(s_.upload(w["xyz_a"], w["cat_a"]), s_.upload(w["xyz_b"], w["cat_b"])) for s_ in sessions]
    cloud_a, cloud_b = clouds[0]
    anchors = torch.from_numpy(w["pairs"]).to(dev)
    p = anchors.shape[0]
    # Two score buffers: the RCCL gather of step k (asynchronous, on RCCL's stream) overlaps the scoring of step k+1.
    outs = [torch.empty(p, dtype=torch.float64, device=dev) for _ in range(2)]
    gathered = [torch.empty(p * world, dtype=torch.float64, device=dev) if (use_dist and rank == 0) else None for _ in range(2)]
    pending = [None, None]
    in_flight = [False, False]
    for s_ in sessions:
        s_.enable_timing(True)
    counter = [0]
    phase_ms = {"cells": 0.0, "anchors": 0.0, "env": 0.0, "sweep": 0.0}
    collect = [False]

    def finish(k):  # wait for the pass enqueued on session k (raises on a device-side error), book its kernel times
        if in_flight[k]:
            sessions[k].finish()
            in_flight[k] = False
            if collect[0]:
                for name, v in sessions[k].last_ms().items():
                    phase_ms[name] += v

    def step():
        k = counter[0] % 2
        counter[0] += 1
        finish(k)  # the session's previous pass (two steps ago) and its score buffer
        if pending[k] is not None:  # the gather that last read this buffer must be done before it is overwritten
            pending[k].wait()
            pending[k] = None
        sessions[k].from_primitives_async(clouds[k][0], clouds[k][1], anchors, w["thr"], outs[k])
        in_flight[k] = True
        if use_dist and args.gather == "step":
            pending[k] = gather_scores(outs[k], gathered[k], world, rank, force_collective=True, async_op=True)

    def drain():
        for k in range(2):
            finish(k)
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for _ in range(args.warmup):
        step()
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    collect[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # every pass and every gather of the timed steps has completed inside the timed region
    collect[0] = False
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    phase_ms = {k: v / max(args.steps, 1) for k, v in phase_ms.items()}

    # Outside the timed region: the same clouds with every anchor used ONCE (pairs (i, i), i < N) -- no environment is
    # shared between pairs, so this is the per-call cost of a plain from_primitives(i, i) call with device-resident inputs.
    extras = {}
    if rank == 0 and args.workload == "c2a" and not args.no_cpu_baseline:  # profiling runs (--no-cpu-baseline) launch only the timed steps
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
        extras = {"unique_anchor_call": {"pairs": int(n_atoms), "ms_per_call": tu * 1e3, "pairs_per_s": n_atoms / tu,
                                         "note": "every anchor used once: no environment re-use between pairs"}}
        sess.from_primitives(cloud_a, cloud_b, anchors, w["thr"], out=outs[(counter[0] - 1) % 2])  # restore for env_points below
    env_points = sess.last_env_points()  # sum over this rank's pairs of n_A + n_B
    out = outs[(counter[0] - 1) % 2]
    scores = out.cpu().numpy()

    if rank == 0:
        algo_bytes = env_points * 28 + 16 * p  # B_pair = (n_A + n_B) * 28 B + 8 B + 8 B
        dom = max(("env", "sweep"), key=lambda k: phase_ms[k])
        dom_name = {"env": "k_env_cells (2 launches: side A + side B)", "sweep": "k_sweep (its launch and the 18 us k_pair_meta record pass in front of it)"}[dom]
        achieved = algo_bytes / (phase_ms[dom] * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(dom_name, w["label"]) if p == 1_000_000 else (None, None)
        result = {
            "metric": "anchor-pair LoCoHD scores/sec",
            "value": p * world * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": w["label"], "pairs_per_gpu": p, "mean_env_points_per_pair": env_points / p,
                       **({"collective": "none in the timed steps (independent anchor pairs); one RCCL gather to rank 0 after them"
                           if args.gather == "end" else "asynchronous RCCL gather of the score slices to rank 0 in every timed step"}
                          if use_dist else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": dom_name,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": phase_ms[dom]},
            "kernel_ms": phase_ms,
            "extras": extras,
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg (and the parity gate on its sample) runs at N = 1 only
            base, err, m = cpu_baseline(w, scores)
            result["cpu_baseline"] = base
            result["max_abs_err_vs_cpu"] = err
            result["parity_sample_pairs"] = m
            if not (err <= 1e-6):
                result["parity_failed"] = True
        final_line = json.dumps(result)
    if use_dist:
        kl = (counter[0] - 1) % 2
        if args.gather == "end":  # one gather of the last step's slices, outside the timed region
            h = gather_scores(outs[kl], gathered[kl], world, rank, force_collective=True, async_op=True)
            if h is not None:
                h.wait()
            torch.cuda.synchronize()
        if rank == 0 and gathered[kl] is not None:
            assert torch.equal(gathered[kl][:p], outs[kl]), "gathered scores differ from the local ones"
        dist.barrier()
        dist.destroy_process_group()
    for s_ in sessions:
        s_.close()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so that the JSON line is the LAST line of stdout
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(final_line, flush=True)
        if json.loads(final_line).get("parity_failed"):
            sys.exit(3)


if __name__ == "__main__":
    main()
