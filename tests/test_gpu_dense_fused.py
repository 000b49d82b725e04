"""Dense rows (from_coords / from_dmxs, src/locohd.rs:410-476) through the fused sort + sweep kernel
(loco_hd_amd/csrc/lchd_dense_fused.hip) and, on the same inputs, through the two-kernel path it replaces
(k_env_rows2 + k_sweep: LCHD_NO_DENSE_FUSED) -- both against the CPU oracle on sampled rows (1e-11) and against each other on
every row.  Row lengths cover one segment (<= 3 072 points per side), two and three, the 10 000-point rows of BASELINE config 2 (four
segments; the two-kernel path runs its 8 193 .. 10 240-point instantiation there) and 20 000-point rows (eight segments)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT = 1e-11
NAMES = [f"c{i}" for i in range(16)]


@pytest.fixture(scope="module")
def lh():
    import loco_hd_amd

    return loco_hd_amd


def cloud(rng, n, n_cat, side):
    return [NAMES[k] for k in rng.integers(0, n_cat, n)], rng.uniform(0.0, side, (n, 3))


def dist_row(x, i):
    d = x[i] - x
    return np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])  # utils.rs:1-8 order


def oracle_rows(lo, sa, sb, rows_a, rows_b, key=None):
    """stat_dist_integral on the stably sorted rows (utils.rs:25-39), one from_anchors call per row pair."""
    out = []
    for ra, rb in zip(rows_a, rows_b):
        ra, rb = np.asarray(ra, dtype=float), np.asarray(rb, dtype=float)
        oa, ob = np.argsort(ra, kind="stable"), np.argsort(rb, kind="stable")
        args = ([sa[k] for k in oa], [sb[k] for k in ob], ra[oa].tolist(), rb[ob].tolist())
        out.append(lo.from_anchors(*args) if key is None else lo.from_anchors(*args, key))
    return np.asarray(out)


def fused_flag(lchd):
    from loco_hd_amd import _native as N

    return int(N.lib().lchd_ctx_last_dense_fused(lchd._context()))


@pytest.mark.parametrize("n,n_cat", [(1025, 5), (2500, 8), (3600, 10), (7000, 12), (10000, 10), (20000, 16)])
def test_from_coords_fused_and_two_kernel_path(lh, oracle, monkeypatch, n, n_cat):
    rng = np.random.default_rng(1000 + n)
    side = (n / 0.05) ** (1 / 3)
    sa, xa = cloud(rng, n, n_cat, side)
    sb, xb = cloud(rng, n, n_cat, side)
    cats, wf = NAMES[:n_cat], ("hyper_exp", [1.0, 0.1])
    fused = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    got = np.asarray(fused.from_coords(sa, sb, xa, xb))
    assert fused_flag(fused) == 1
    assert np.all(np.isfinite(got)) and got.min() >= 0.0 and got.max() <= 1.0
    again = np.asarray(fused.from_coords(sa, sb, xa, xb))
    assert np.array_equal(got, again)  # same call, same bits
    rows = sorted(set(rng.integers(0, n, 24).tolist()) | {0, n - 1})
    lo = oracle.LoCoHD(cats, oracle.WeightFunction(*wf))
    want = oracle_rows(lo, sa, sb, [dist_row(xa, i) for i in rows], [dist_row(xb, i) for i in rows])
    assert np.max(np.abs(got[rows] - want)) < TIGHT
    # the two-kernel path (row sort to the environment store, then the sweep) on the same input
    monkeypatch.setenv("LCHD_NO_DENSE_FUSED", "1")
    plain = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    old = np.asarray(plain.from_coords(sa, sb, xa, xb))
    assert fused_flag(plain) == 0
    assert np.max(np.abs(old[rows] - want)) < TIGHT
    assert np.max(np.abs(old - got)) < 1e-13


def test_from_dmxs_ten_thousand_point_rows(lh, oracle, monkeypatch):
    """Given distance rows of 10 000 / 9 000 points (unequal widths), +inf entries (python_codes/ensembles/
    compare_ensembles.py:261-263), through both paths."""
    rng = np.random.default_rng(77)
    na, nb, rows = 10000, 9000, 40
    sa, xa = cloud(rng, na, 10, 58.0)
    sb, xb = cloud(rng, nb, 10, 58.0)
    da = np.stack([dist_row(xa, i) for i in range(rows)])
    db = np.stack([dist_row(xb, i) for i in range(rows)])
    da[rng.integers(0, rows, 60), rng.integers(rows, na, 60)] = np.inf
    db[rng.integers(0, rows, 30), rng.integers(rows, nb, 30)] = np.inf
    cats, wf = NAMES[:10], ("uniform", [3.0, 40.0])
    lo = oracle.LoCoHD(cats, oracle.WeightFunction(*wf))
    want = oracle_rows(lo, sa, sb, da, db)
    fused = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    got = np.asarray(fused.from_dmxs(sa, sb, da, db))
    assert fused_flag(fused) == 1
    assert np.max(np.abs(got - want)) < TIGHT
    monkeypatch.setenv("LCHD_NO_DENSE_FUSED", "1")
    plain = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    old = np.asarray(plain.from_dmxs(sa, sb, da, db))
    assert fused_flag(plain) == 0
    assert np.max(np.abs(old - want)) < TIGHT


def test_ragged_rows_and_weight_function_dictionary(lh, oracle):
    """Rows of different lengths (utils.rs:25-39) with a per-row weight function (src/locohd.rs:230-283), every family."""
    rng = np.random.default_rng(78)
    n, rows = 6000, 48
    s, x = cloud(rng, n, 7, 45.0)
    full = np.stack([dist_row(x, i) for i in range(rows)])
    lens_a, lens_b = rng.integers(1500, n + 1, rows), rng.integers(1100, n + 1, rows)
    ra = [full[i, : int(lens_a[i])].tolist() for i in range(rows)]
    rb = [full[(i * 5) % rows, : int(lens_b[i])].tolist() for i in range(rows)]
    for r in rb:
        k = int(np.argmin(r))
        r[0], r[k] = 0.0, r[0]  # the anchor (distance 0) must be in the row
    wfs = {"he": ("hyper_exp", [1.0, 2.0, 0.5, 0.1]), "un": ("uniform", [2.0, 30.0]), "da": ("dagum", [2.5, 12.0, 0.8]),
           "ku": ("kumaraswamy", [1.0, 35.0, 2.0, 3.0])}
    keys = [list(wfs)[k] for k in rng.integers(0, 4, rows)]
    cats = NAMES[:7]
    lchd = lh.LoCoHD(cats, {k: lh.WeightFunction(*v) for k, v in wfs.items()})
    got = np.asarray(lchd.from_dmxs(s, s, ra, rb, keys))
    assert fused_flag(lchd) == 1
    lo = oracle.LoCoHD(cats, {k: oracle.WeightFunction(*v) for k, v in wfs.items()})
    want = np.asarray([oracle_rows(lo, s, s, [ra[i]], [rb[i]], keys[i])[0] for i in range(rows)])
    assert np.max(np.abs(got - want)) < TIGHT


def test_lattice_ties_and_identical_structures(lh, oracle):
    """Thousands of exactly equal distances: buckets of the fused sort overflow and the call is repeated by the two-kernel path;
    identical structures score exactly 0 either way."""
    grid = np.array(list(itertools.product(range(12), repeat=3)), dtype=float)  # 1728 lattice points
    rng = np.random.default_rng(79)
    sa = [NAMES[k] for k in rng.integers(0, 6, len(grid))]
    sb = [NAMES[k] for k in rng.integers(0, 6, len(grid))]
    cats, wf = NAMES[:6], ("hyper_exp", [1.0, 0.25])
    lchd = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    got = np.asarray(lchd.from_coords(sa, sb, grid, grid[::-1].copy()))
    rows = list(range(0, len(grid), 61))
    lo = oracle.LoCoHD(cats, oracle.WeightFunction(*wf))
    want = oracle_rows(lo, sa, sb, [dist_row(grid, i) for i in rows], [dist_row(grid[::-1], i) for i in rows])
    assert np.max(np.abs(got[rows] - want)) < TIGHT
    n = 3000
    s, x = cloud(rng, n, 6, 39.0)
    same = np.asarray(lchd.from_coords(s, s, x, x))
    assert fused_flag(lchd) == 1
    assert np.max(np.abs(same)) == 0.0
    # duplicates of the anchor's coordinates (several points at distance exactly 0) and a structure far from the origin
    x2 = x + 1.0e4
    x2[5] = x2[17] = x2[0]
    got2 = np.asarray(lchd.from_coords(s, s[::-1], x2, x2[::-1].copy()))
    rows = [0, 5, 17, 1234, n - 1]
    want2 = oracle_rows(lo, s, s[::-1], [dist_row(x2, i) for i in rows], [dist_row(x2[::-1], i) for i in rows])
    assert np.max(np.abs(got2[rows] - want2)) < TIGHT


def test_a_sample_that_misrepresents_the_rows(lh, oracle, monkeypatch):
    """The fused kernel plans a row pair's distance segments from every 4th point (lchd_dense_fused.hip).  Here every 4th atom lies in a
    far-away cluster and the others in a compact one, so for the rows whose sample is that residue class the plan is wrong by thousands of
    events: the segment outgrows its LDS arrays, the kernel reports the row (ST_ROW_RETRY) and the host repeats the call with the
    two-kernel path.  Same scores either way; the ordinary random cloud of the same size stays on the fused path."""
    rng = np.random.default_rng(4242)
    n, n_cat = 5000, 9
    cats, wf = NAMES[:n_cat], ("hyper_exp", [1.0, 0.1])

    def skewed():
        x = rng.uniform(0.0, 30.0, (n, 3))
        x[::4] += 400.0  # atoms 0, 4, 8, ...: a second cluster 700 A away
        return [NAMES[k] for k in rng.integers(0, n_cat, n)], x

    (sa, xa), (sb, xb) = skewed(), skewed()
    lchd = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    got = np.asarray(lchd.from_coords(sa, sb, xa, xb))
    assert fused_flag(lchd) == 0  # the call was repeated by the two-kernel path
    rows = sorted(set(rng.integers(0, n, 12).tolist()) | {0, 1, 2, 3, 4, n - 1})
    lo = oracle.LoCoHD(cats, oracle.WeightFunction(*wf))
    want = oracle_rows(lo, sa, sb, [dist_row(xa, i) for i in rows], [dist_row(xb, i) for i in rows])
    assert np.max(np.abs(got[rows] - want)) < TIGHT
    monkeypatch.setenv("LCHD_NO_DENSE_FUSED", "1")
    plain = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    assert np.array_equal(np.asarray(plain.from_coords(sa, sb, xa, xb)), got)
    monkeypatch.delenv("LCHD_NO_DENSE_FUSED")
    s, x = cloud(rng, n, n_cat, 46.0)
    t, y = cloud(rng, n, n_cat, 46.0)
    fused = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    ok = np.asarray(fused.from_coords(s, t, x, y))
    assert fused_flag(fused) == 1
    rows = [0, 77, n - 1]
    assert np.max(np.abs(ok[rows] - oracle_rows(lo, s, t, [dist_row(x, i) for i in rows], [dist_row(y, i) for i in rows]))) < TIGHT


def test_random_shapes_against_the_two_kernel_path(lh, oracle, monkeypatch):
    """A sweep over row lengths (one to four distance segments, sizes next to the LDS arrays' capacity of 6144 events per segment: 3072
    points per side), category counts (all three instantiations), weight functions and box shapes: every row of the fused kernel against the
    two-kernel path (1e-13), two rows per case against the oracle."""
    rng = np.random.default_rng(555)
    wfs = [("hyper_exp", [1.0, 0.1]), ("hyper_exp", [0.6, 0.3, 0.1, 0.25, 0.1, 0.04]), ("uniform", [3.0, 25.0]), ("dagum", [2.5, 12.0, 0.8]),
           ("hyper_exp", [0.25, 0.25, 0.25, 0.25, 1.0, 0.5, 0.2, 0.05])]
    cases = [(3071, 8), (3072, 5), (3073, 12), (4100, 16), (1030, 3), (6100, 9), (6200, 11), (8191, 7), (2048, 16), (5000, 1)]
    for k, (n, n_cat) in enumerate(cases):
        wf = wfs[k % len(wfs)]
        box = np.array([1.0, 1.0 + (k % 3), 1.0 + (k % 2) * 4.0])  # cubes, slabs and rods: different distance distributions
        scale = (n / 0.05 / box.prod()) ** (1 / 3)
        sa, xa = [NAMES[c] for c in rng.integers(0, n_cat, n)], rng.uniform(0.0, 1.0, (n, 3)) * box * scale
        sb, xb = [NAMES[c] for c in rng.integers(0, n_cat, n)], rng.uniform(0.0, 1.0, (n, 3)) * box * scale
        cats = NAMES[:n_cat]
        monkeypatch.delenv("LCHD_NO_DENSE_FUSED", raising=False)
        fused = lh.LoCoHD(cats, lh.WeightFunction(*wf))
        got = np.asarray(fused.from_coords(sa, sb, xa, xb))
        assert fused_flag(fused) == 1, (n, n_cat)
        monkeypatch.setenv("LCHD_NO_DENSE_FUSED", "1")
        plain = lh.LoCoHD(cats, lh.WeightFunction(*wf))
        old = np.asarray(plain.from_coords(sa, sb, xa, xb))
        assert fused_flag(plain) == 0
        assert np.max(np.abs(old - got)) < 1e-13, (n, n_cat, wf)
        rows = [int(rng.integers(0, n)), n - 1]
        lo = oracle.LoCoHD(cats, oracle.WeightFunction(*wf))
        want = oracle_rows(lo, sa, sb, [dist_row(xa, i) for i in rows], [dist_row(xb, i) for i in rows])
        assert np.max(np.abs(got[rows] - want)) < TIGHT, (n, n_cat, wf)


def test_other_configurations_keep_the_two_kernel_path(lh, oracle):
    """Category weights, another statistical distance or more than 16 categories: not the fused kernel's configuration."""
    rng = np.random.default_rng(80)
    n = 2000
    s, x = cloud(rng, n, 5, 34.0)
    t, y = cloud(rng, n, 5, 34.0)
    rows = [0, 7, 999, n - 1]
    ra, rb = [dist_row(x, i) for i in rows], [dist_row(y, i) for i in rows]
    for kw in ({"category_weights": [1.0, 0.5, 2.0, 3.0, 0.25]}, {"statistical_distance": ("Kolmogorov-Smirnov", [])}):
        def build(mod):
            args = dict(kw)
            if "statistical_distance" in args:
                args["statistical_distance"] = mod.StatisticalDistance(*args["statistical_distance"])
            return mod.LoCoHD(NAMES[:5], mod.WeightFunction("hyper_exp", [1.0, 0.2]), **args)

        lchd = build(lh)
        got = np.asarray(lchd.from_coords(s, t, x, y))
        assert fused_flag(lchd) == 0
        assert np.max(np.abs(got[rows] - oracle_rows(build(oracle), s, t, ra, rb))) < TIGHT


def test_error_behaviour(lh):
    rng = np.random.default_rng(81)
    n = 1500
    s, x = cloud(rng, n, 4, 30.0)
    lchd = lh.LoCoHD(NAMES[:4], lh.WeightFunction("uniform", [3.0, 10.0]))
    bad = list(s)
    bad[1400] = "unknown"
    with pytest.raises(ValueError):  # pmf.rs:38-42
        lchd.from_coords(bad, s, x, x)
    d = np.stack([dist_row(x, i) for i in range(8)])
    d2 = d.copy()
    d2[3, 3] = 0.5  # no distance of 0 in the row: src/locohd.rs:74-77
    with pytest.raises(ValueError):
        lchd.from_dmxs(s, s, d, d2)
    d3 = d.copy()
    d3[2, 700] = -1.0
    with pytest.raises(ValueError):
        lchd.from_dmxs(s, s, d3, d)
    ok = np.asarray(lchd.from_dmxs(s, s, d, d))  # the context still works afterwards
    assert fused_flag(lchd) == 1 and np.max(np.abs(ok)) == 0.0
