"""C2a-style sweep against cloud size: N atoms per cloud (0.05 atoms / A^3), 10 categories, 10^6 pairs = permutation rounds over all atoms.
Prints ms per call (sweep-dominated) for the library in LCHD_LIB, with and without prefix-count rows (LCHD_PRE_ROWS=-1)."""
import sys, time, json, os
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))  # repository root
import loco_hd_amd as lh
from loco_hd_amd.device import DeviceSession
res = {}
for n in [int(x) for x in sys.argv[1:]] or [10000, 40000]:
    rng = np.random.default_rng(2)
    side = (n / 0.05) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, 10, n).astype(np.int32), rng.integers(0, 10, n).astype(np.int32)
    rounds = max(1, 1_000_000 // n)
    pairs = np.concatenate([np.stack([rng.permutation(n), rng.permutation(n)], 1) for _ in range(rounds)])
    l = lh.LoCoHD([f"c{i}" for i in range(10)], lh.WeightFunction("hyper_exp", [1.0, 0.1]))
    s = DeviceSession(l); s.enable_timing(True)
    a, b = s.upload(xa, ca), s.upload(xb, cb)
    an = torch.from_numpy(pairs).cuda()
    out = torch.empty(len(pairs), dtype=torch.float64, device="cuda")
    for _ in range(6): s.from_primitives(a, b, an, 10.0, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): s.from_primitives(a, b, an, 10.0, out=out)
    torch.cuda.synchronize()
    res[f"n{n}_pairs{len(pairs)}_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
    res[f"n{n}_phase_ms"] = {k: round(v, 4) for k, v in s.last_ms().items()} if isinstance(s.last_ms(), dict) else s.last_ms()
    s.close()
print(os.environ.get("LCHD_LIB", "shipped").split("/")[-1], os.environ.get("LCHD_PRE_ROWS", ""), json.dumps(res))
