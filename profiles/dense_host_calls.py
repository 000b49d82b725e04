"""dense from_coords timing: end-to-end host call + kernel phases.  LCHD_OLD_ROWS=1 selects the old row kernel."""
import json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))  # repository root
import loco_hd_amd as lh
from loco_hd_amd import _native as N
res = {}
names = [f"c{i}" for i in range(10)]
for nn in (3000, 10_000, 16_000, 20_000):
    rng = np.random.default_rng(2)
    side = (nn / 0.05) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (nn, 3)), rng.uniform(0, side, (nn, 3))
    sa, sb = [names[i] for i in rng.integers(0, 10, nn)], [names[i] for i in rng.integers(0, 10, nn)]
    l3 = lh.LoCoHD(names, lh.WeightFunction("hyper_exp", [1.0, 0.1]))
    out0 = l3.from_coords(sa, sb, xa, xb)
    N.lib().lchd_ctx_enable_timing(l3._context(), 1)
    reps = 3 if nn >= 10_000 else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        out = l3.from_coords(sa, sb, xa, xb)
    t = (time.perf_counter() - t0) / reps * 1e3
    res[nn] = {"ms_host_call": t, "dense_pairs_per_s": nn / t * 1e3, "env_ms": N.lib().lchd_ctx_last_ms(l3._context(), b"env"),
               "sweep_ms": N.lib().lchd_ctx_last_ms(l3._context(), b"sweep"), "checksum": float(np.sum(out)), "same": bool(out == out0)}
print(json.dumps(res))
