"""How many candidate SLOTS (groups of 8 records) does k_env_group's row-run search touch per anchor with cells of thr / 2 (reach 2) against
thr / 3 (reach 3)?  Uniform random cloud of C4's density (0.023 atoms / A^3) and C5's (0.05), thr 10."""
import numpy as np
def slots(n, dens, thr, reach, seed=0, n_anchor=400):
    rng = np.random.default_rng(seed)
    L = (n / dens) ** (1 / 3)
    x = rng.uniform(0, L, (n, 3))
    cell = thr / reach * (1 + 1e-9)
    dim = max(int(L // cell), 1)
    cs = L / dim if dim * cell > L else cell
    cs = L / dim
    cid = np.minimum((x / cs).astype(int), dim - 1)
    # count per (z, y, x) cell
    cnt = np.zeros((dim, dim, dim), dtype=int)
    np.add.at(cnt, (cid[:, 2], cid[:, 1], cid[:, 0]), 1)
    cum = np.concatenate([np.zeros((dim, dim, 1), int), np.cumsum(cnt, axis=2)], axis=2)
    tot_slots = tot_cand = tot_kept = 0
    for a in rng.integers(0, n, n_anchor):
        p = x[a]; c = cid[a]; f = p / cs - c
        kept = (((x - p) ** 2).sum(1) < thr * thr).sum()
        for oz in range(-reach, reach + 1):
            for oy in range(-reach, reach + 1):
                zz, yy = c[2] + oz, c[1] + oy
                if not (0 <= zz < dim and 0 <= yy < dim): continue
                def gap(o, fr): return 0.0 if o == 0 else max((fr - (o + 1)) if o < 0 else ((1 - fr) + (o - 1)), 0.0) * cs
                gy, gz = gap(oy, f[1]), gap(oz, f[2])
                r2 = gy * gy + gz * gz
                if r2 >= thr * thr: continue
                lo = hi = 0
                for k in range(reach, 0, -1):
                    if r2 + ((max(f[0], 0) + (k - 1)) * cs) ** 2 < thr * thr: lo = -k; break
                for k in range(reach, 0, -1):
                    if r2 + ((max(1 - f[0], 0) + (k - 1)) * cs) ** 2 < thr * thr: hi = k; break
                xl, xh = max(c[0] + lo, 0), min(c[0] + hi, dim - 1)
                ln = cum[zz, yy, xh + 1] - cum[zz, yy, xl]
                tot_cand += ln; tot_slots += (ln + 7) // 8 * 8
        tot_kept += kept
    return tot_slots / n_anchor, tot_cand / n_anchor, tot_kept / n_anchor
for name, n, dens in (("C4-like", 60000, 0.023), ("C5-like", 60000, 0.05)):
    for reach in (2, 3, 4):
        s, c, k = slots(n, dens, 10.0, reach)
        print(f"{name} cells thr/{reach}: slots {s:7.1f}  candidates {c:7.1f}  kept {k:6.1f}  slots/kept {s / k:5.2f}")
