"""The two small device calls of VERDICT r05 item 2b, for a kernel-trace timeline: 10^4 unique anchors and a rank's 125 000-pair share of C2a."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))  # repository root
import bench
import loco_hd_amd as lh
from loco_hd_amd.device import DeviceSession
w = bench.make_workload("c2a", 0, 1_000_000)
l2 = lh.LoCoHD([f"c{i}" for i in range(w["C"])], lh.WeightFunction(*w["wf"]))
s2 = DeviceSession(l2)
a, b = s2.upload(w["xyz_a"], w["cat_a"]), s2.upload(w["xyz_b"], w["cat_b"])
anchors = torch.from_numpy(w["pairs"]).cuda()
out = torch.empty(len(w["pairs"]), dtype=torch.float64, device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
def timed(fn, reps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
if which in ("both", "unique"):
    print("unique_anchor_call_ms", timed(lambda: s2.from_primitives(a, b, anchors[:10000], w["thr"], out=out), 50, 10))
if which in ("both", "125k"):
    print("c2a_125k_ms", timed(lambda: s2.from_primitives(a, b, anchors[:125000], w["thr"], out=out), 30, 10))
