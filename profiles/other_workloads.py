#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md section 5 (run on the MI355X box from the repository root):
    python3 profiles/other_workloads.py > gpurun_out/other_workloads.json
C3 all-vs-all batch, C2a with other distances / weights, C2b dense from_coords, host-pointer calls, 10^7 pairs per call.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import loco_hd_amd as lh  # noqa: E402
from loco_hd_amd.device import DeviceSession  # noqa: E402

res = {}


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


# ---- C3: 50 decoys x 3000 points, every 3rd point a "Cent" anchor, all unordered decoy pairs, accept_same=False ----------
rng = np.random.default_rng(3)
types = ["Cent", "AmideC", "OH", "Pos", "Neg", "Aro", "Ali", "Sulf"]
n, nd = 3000, 50
side = (n / 0.023) ** (1 / 3)
base = rng.uniform(0, side, (n, 3))
cat = np.where(np.arange(n) % 3 == 0, 0, rng.integers(1, 8, n)).astype(np.int32)
tag = (np.arange(n) // 3).astype(np.int32)
lchd = lh.LoCoHD(types, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule({"accept_same": False}))
sess = DeviceSession(lchd)
batch, offs = sess.upload_batch([(base + rng.normal(0, 1.5, base.shape), cat, tag) for _ in range(nd)])
la = np.arange(0, n, 3)
pairs = np.concatenate([np.stack([offs[a] + la, offs[b] + la], 1) for a in range(nd) for b in range(a + 1, nd)])
anchors = torch.from_numpy(pairs).cuda()
out = torch.empty(len(pairs), dtype=torch.float64, device="cuda")
sess.enable_timing(True)
ms = timed(lambda: sess.from_primitives(batch, batch, anchors, 10.0, out=out))
res["c3_all_vs_all"] = {"pairs": len(pairs), "ms_per_call": ms, "pairs_per_s": len(pairs) / ms * 1e3, "kernel_ms": sess.last_ms()}
sess.close()

# ---- one structure pair per call (what a reference user does): device-resident inputs, latency per call -----------------
for nn in (1000, 3000):
    rs = np.random.default_rng(0)
    sd = (nn / 0.023) ** (1 / 3)
    xa, xb = rs.uniform(0, sd, (nn, 3)), rs.uniform(0, sd, (nn, 3))
    ct = rs.integers(0, 8, nn).astype(np.int32)
    tg = (np.arange(nn) // 3).astype(np.int32)
    s1 = DeviceSession(lchd)
    ha, hb = s1.upload(xa, ct, tg), s1.upload(xb, ct, tg)
    an = torch.from_numpy(np.stack([np.arange(0, nn, 3), np.arange(0, nn, 3)], 1)).cuda()
    o1 = torch.empty(len(an), dtype=torch.float64, device="cuda")
    t = timed(lambda: s1.from_primitives(ha, hb, an, 10.0, out=o1), reps=200, warm=20)  # (steady state: the first calls after start-up run at idle clocks)
    res[f"single_structure_pair_{nn}_atoms"] = {"pairs": len(an), "ms_per_call": t}
    s1.close()

# ---- the reference's own call: LoCoHD.from_primitives(list[PrimitiveAtom], list[PrimitiveAtom], anchors, threshold) --------
nn = 3000
rs = np.random.default_rng(0)
sd = (nn / 0.023) ** (1 / 3)
pa = [lh.PrimitiveAtom(types[i % 8], f"A/{i // 3}-RES", rs.uniform(0, sd, 3)) for i in range(nn)]
pb = [lh.PrimitiveAtom(types[i % 8], f"A/{i // 3}-RES", rs.uniform(0, sd, 3)) for i in range(nn)]
anc = [(i, i) for i in range(0, nn, 3)]
t = timed(lambda: lchd.from_primitives(pa, pb, anc, 10.0), reps=20, warm=3)  # the same lists again and again: served from the packed-list cache
t_first = timed(lambda: lchd.from_primitives(list(pa), list(pb), list(anc), 10.0), reps=20, warm=3)  # new list objects every call: full extraction
pb_k = [[lh.PrimitiveAtom(types[i % 8], f"A/{i // 3}-RES", rs.uniform(0, sd, 3)) for i in range(nn)] for _ in range(8)]
kk = [0]
def one_vs_many():  # the casp14 loop: one native structure against another decoy every call (side A cached, side B extracted)
    kk[0] += 1
    return lchd.from_primitives(pa, list(pb_k[kk[0] % 8]), anc, 10.0)
t_mixed = timed(one_vs_many, reps=24, warm=3)
it = {}
ka, kb = lchd.pack(pa, it), lchd.pack(pb, it)
t2 = timed(lambda: lchd.from_packed(ka, kb, np.asarray(anc, dtype=np.int64), 10.0, interner=it), reps=50, warm=3)
res["reference_style_call_3000_atoms_1000_pairs"] = {"from_primitives_lists_ms": t, "from_primitives_new_lists_every_call_ms": t_first,
                                                     "from_primitives_same_a_new_b_ms": t_mixed, "c_abi_host_pointer_call_ms": t2}

# ---- C2a clouds with other configurations (sweep phase per 10^6 pairs) ------------------------------------------------
w = bench.make_workload("c2a", 0, 1_000_000)
names = [f"c{i}" for i in range(w["C"])]
anchors = torch.from_numpy(w["pairs"]).cuda()
out = torch.empty(len(w["pairs"]), dtype=torch.float64, device="cuda")
configs = {
    "hellinger2_default": dict(),
    "hellinger2_category_weights": dict(category_weights=list(np.linspace(0.5, 2.0, w["C"]))),
    "two_weight_functions": dict(w_func={"a": lh.WeightFunction("hyper_exp", [1.0, 0.1]), "b": lh.WeightFunction("uniform", [3.0, 10.0])}),
    "kolmogorov_smirnov": dict(statistical_distance=lh.StatisticalDistance("Kolmogorov-Smirnov", [])),
    "kullback_leibler": dict(statistical_distance=lh.StatisticalDistance("Kullback-Leibler", [1e-10])),  # eps <= 1e-9: k_sweep_inc, O(1) per event
    "renyi_2.4": dict(statistical_distance=lh.StatisticalDistance("Renyi", [2.4, 1e-10])),
    "renyi_0.5": dict(statistical_distance=lh.StatisticalDistance("Renyi", [0.5, 1e-10])),
    "kullback_leibler_eps_1e-3": dict(statistical_distance=lh.StatisticalDistance("Kullback-Leibler", [1e-3])),  # the generic per-category sweep
    "renyi_2.4_eps_1e-3": dict(statistical_distance=lh.StatisticalDistance("Renyi", [2.4, 1e-3])),
    "hellinger_exponent_3": dict(statistical_distance=lh.StatisticalDistance("Hellinger", [3.0])),
    "hellinger_exponent_2.5": dict(statistical_distance=lh.StatisticalDistance("Hellinger", [2.5])),
}
res["c2a_variants_sweep_ms"] = {}
for k, kw in configs.items():
    kw = dict(kw)
    wf = kw.pop("w_func", lh.WeightFunction(*w["wf"]))
    l2 = lh.LoCoHD(names, wf, **kw)
    s2 = DeviceSession(l2)
    s2.enable_timing(True)
    a, b = s2.upload(w["xyz_a"], w["cat_a"]), s2.upload(w["xyz_b"], w["cat_b"])
    wfi = None
    if isinstance(wf, dict):
        wfi = torch.from_numpy((np.arange(len(w["pairs"])) % 2).astype(np.int32)).cuda()
    reps = 1 if k in ("renyi_2.4", "hellinger_exponent_3") else 3
    timed(lambda: s2.from_primitives(a, b, anchors, w["thr"], out=out, wf_index=wfi), reps=reps, warm=1)
    res["c2a_variants_sweep_ms"][k] = s2.last_ms()["sweep"]
    s2.close()

# ---- 10^7 pairs in one device call ------------------------------------------------------------------------------------
l2 = lh.LoCoHD(names, lh.WeightFunction(*w["wf"]))
s2 = DeviceSession(l2)
a, b = s2.upload(w["xyz_a"], w["cat_a"]), s2.upload(w["xyz_b"], w["cat_b"])
big = anchors.repeat(10, 1).contiguous()
obig = torch.empty(len(big), dtype=torch.float64, device="cuda")
ms = timed(lambda: s2.from_primitives(a, b, big, w["thr"], out=obig), reps=3, warm=1)
s2.from_primitives(a, b, anchors, w["thr"], out=out)
res["ten_million_pairs_one_call"] = {"ms": ms, "pairs_per_s": len(big) / ms * 1e3, "bitwise_equal_to_1e6_calls": bool(torch.equal(obig[: len(out)], out))}
s2.close()

# ---- host-pointer API (lchd_from_primitives: packing + H2D + D2H inside the call) ------------------------------------------
pa, pb = lh.api._Packed(w["xyz_a"], w["cat_a"], np.zeros(w["n"], np.int32)), lh.api._Packed(w["xyz_b"], w["cat_b"], np.zeros(w["n"], np.int32))
for m in (10_000, 1_000_000):
    t = timed(lambda: l2.from_packed(pa, pb, w["pairs"][:m], w["thr"]), reps=6, warm=3)
    res[f"host_pointer_call_{m}_pairs"] = {"ms": t, "pairs_per_s": m / t * 1e3}

# ---- C2b dense from_coords --------------------------------------------------------------------------------------------
for nn in (10_000, 20_000):
    rng = np.random.default_rng(2)
    xa, xb = rng.uniform(0, 58.5, (nn, 3)), rng.uniform(0, 58.5, (nn, 3))
    sa, sb = [names[i] for i in rng.integers(0, 10, nn)], [names[i] for i in rng.integers(0, 10, nn)]
    l3 = lh.LoCoHD(names, lh.WeightFunction(*w["wf"]))
    l3.from_coords(sa, sb, xa, xb)  # first call at this size: grows the context's workspace (a multi-GB hipMalloc)
    t0 = time.perf_counter()
    l3.from_coords(sa, sb, xa, xb)
    t = (time.perf_counter() - t0) * 1e3
    res[f"c2b_dense_from_coords_{nn}"] = {"ms_end_to_end_host_call": t, "dense_pairs_per_s": nn / t * 1e3,
                                          "algorithmic_GB_per_s": nn * (2 * nn * 28 + 16) / t / 1e6}
print(json.dumps(res, indent=1))
