"""A/B of call latencies: LCHD_LIB selects the library.  Prints one JSON line."""
import json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))  # repository root
import bench
import loco_hd_amd as lh
from loco_hd_amd.device import DeviceSession

def timed(fn, reps=200, warm=20, stats=None, name=None):
    """Mean over `reps` back-to-back calls.  With `stats`: every call is timed on its own (synchronous calls only) and the
    median / max go into stats[name + "_median_ms" / "_max_ms"] -- round 3's host_ptr_call_1000_atoms figure (0.47 ms) was a MEAN
    over 100 calls of which ONE, the first measured series of the process, took 40 ms (a one-off of the runtime: the same call
    measured after a 3000-atom series shows no such stall; median 0.071 ms, one pass per call either way)."""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    if stats is not None:
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        ts = np.asarray(ts) * 1e3
        stats[name + "_median_ms"] = float(np.median(ts)); stats[name + "_max_ms"] = float(ts.max())
        return float(ts.mean())
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
res = {}
types = ["Cent", "AmideC", "OH", "Pos", "Neg", "Aro", "Ali", "Sulf"]
lchd = lh.LoCoHD(types, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule({"accept_same": False}))
for nn in (1000, 3000):
    rs = np.random.default_rng(0)
    sd = (nn / 0.023) ** (1 / 3)
    xa, xb = rs.uniform(0, sd, (nn, 3)), rs.uniform(0, sd, (nn, 3))
    ct = rs.integers(0, 8, nn).astype(np.int32); tg = (np.arange(nn) // 3).astype(np.int32)
    s1 = DeviceSession(lchd)
    ha, hb = s1.upload(xa, ct, tg), s1.upload(xb, ct, tg)
    an = torch.from_numpy(np.stack([np.arange(0, nn, 3), np.arange(0, nn, 3)], 1)).cuda()
    o1 = torch.empty(len(an), dtype=torch.float64, device="cuda")
    res[f"dev_call_{nn}_atoms_ms"] = timed(lambda: s1.from_primitives(ha, hb, an, 10.0, out=o1))
    # enqueue-only share: time from call to return of the async form, then the wait
    te = tw = 0.0
    for _ in range(100):
        t0 = time.perf_counter(); s1.from_primitives_async(ha, hb, an, 10.0, o1); t1 = time.perf_counter(); s1.finish(); t2 = time.perf_counter()
        te += t1 - t0; tw += t2 - t1
    res[f"dev_call_{nn}_enqueue_ms"] = te / 100 * 1e3; res[f"dev_call_{nn}_wait_ms"] = tw / 100 * 1e3
    it = {}
    pa = lh.api._Packed(xa, ct, tg); pb = lh.api._Packed(xb, ct, tg)
    anh = an.cpu().numpy()
    p0 = lh._native.lib().lchd_ctx_pass_count(lchd._context())
    res[f"host_ptr_call_{nn}_atoms_ms"] = timed(lambda: lchd.from_packed(pa, pb, anh, 10.0), reps=100, stats=res, name=f"host_ptr_call_{nn}_atoms")
    res[f"host_ptr_call_{nn}_passes_per_call"] = (lh._native.lib().lchd_ctx_pass_count(lchd._context()) - p0) / 120
    s1.close()
w = bench.make_workload("c2a", 0, 1_000_000)
l2 = lh.LoCoHD([f"c{i}" for i in range(w["C"])], lh.WeightFunction(*w["wf"]))
s2 = DeviceSession(l2); s2.enable_timing(True)
a, b = s2.upload(w["xyz_a"], w["cat_a"]), s2.upload(w["xyz_b"], w["cat_b"])
anchors = torch.from_numpy(w["pairs"]).cuda()
out = torch.empty(len(w["pairs"]), dtype=torch.float64, device="cuda")
res["c2a_unique_anchor_call_ms"] = timed(lambda: s2.from_primitives(a, b, anchors[:10000], w["thr"], out=out), reps=50, warm=5)
res["c2a_unique_phase_ms"] = s2.last_ms()
res["c2a_1e6_ms"] = timed(lambda: s2.from_primitives(a, b, anchors, w["thr"], out=out), reps=10, warm=3)
res["c2a_1e6_phase_ms"] = s2.last_ms()
res["c2a_125k_ms"] = timed(lambda: s2.from_primitives(a, b, anchors[:125000], w["thr"], out=out), reps=20, warm=3)
res["c2a_125k_phase_ms"] = s2.last_ms()
lf = lh.LoCoHD(["O", "A", "B", "C"], lh.WeightFunction("uniform", [0.0, 4.0]))
seq = ["O", "A", "B", "C"]
res["from_anchors_call_ms"] = timed(lambda: lf.from_anchors(seq, seq, [0.0, 1.0, 2.0, 3.0], [0.0, 1.0, 1.0, 1.0]), reps=100)
print(json.dumps(res))
