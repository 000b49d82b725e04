"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's known answers.

Bar (BASELINE.json north_star): anchor indexing bit-exact (output i belongs to anchor pair i), scores
within 1e-6 absolute.  In practice the two paths agree to ~1e-14; TOL below is the contractual bound,
TIGHT the regression bound.
"""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6
TIGHT = 1e-11

CATS = ["A", "B", "C", "D", "E"]


@pytest.fixture(scope="module")
def lh():
    import loco_hd_amd

    return loco_hd_amd


def both(lh, oracle, build):
    """build(mod) -> value; returns (hip, oracle) results"""
    return build(lh), build(oracle)


def cloud(rng, n, box=50.0, cats=CATS):
    return rng.choice(cats, n).tolist(), rng.uniform(-box, box, (n, 3))


def prims(mod, seq, xyz, tags=None):
    tags = [""] * len(seq) if tags is None else tags
    return [mod.PrimitiveAtom(s, t, c) for s, t, c in zip(seq, tags, xyz)]


# ---- the reference's own known answers, through the GPU ------------------------------------------------
def test_small_locohd(lh):  # /root/reference/tests/test_locohd.py:27-52
    lchd = lh.LoCoHD(["O", "A", "B", "C"], lh.WeightFunction("uniform", [0.0, 4.0]))
    seq = ["O", "A", "B", "C"]
    assert lchd.from_anchors(seq, seq, [0.0, 1.0, 2.0, 3.0], [0.0, 1.0, 1.0, 1.0]) == pytest.approx(0.2268, abs=5e-5)
    assert lchd.from_anchors(seq, seq, [0.0, 1.0, 1.0, 1.0], [0.0, 1.0, 2.0, 3.0]) == pytest.approx(0.2268, abs=5e-5)
    lchd = lh.LoCoHD(["A", "B", "C"], lh.WeightFunction("kumaraswamy", [3.0, 10.0, 2.0, 5.0]))
    v = lchd.from_anchors(["A", "B", "A", "C"], ["A", "C"], [0.0, 1.0, 5.0, 9.0], [0.0, 7.0])
    assert v == pytest.approx(0.4979, abs=5e-5)
    assert v == pytest.approx(0.4978635773685092, abs=TIGHT)


def test_tag_rule_in_locohd(lh):  # /root/reference/tests/test_tag_pairing_rule.py:100-157
    from test_oracle_kat import PLANAR_ANCHORS, planar_structure

    s = planar_structure(lh)
    wf = lh.WeightFunction("uniform", [1.0, 1.001])
    lchd = lh.LoCoHD(["A", "B", "C"], wf, lh.TagPairingRule({"accept_same": True}))
    got = lchd.from_primitives(s, s, PLANAR_ANCHORS, 1.002)
    for g, w in zip(got, [0.0, 0.0, 1.0, 1.0, 1.0]):
        assert abs(g - w) < 5e-16
    lchd = lh.LoCoHD(["A", "B", "C"], wf, lh.TagPairingRule({"accept_same": False}))
    got = lchd.from_primitives(s, s, PLANAR_ANCHORS, 1.002)
    for g, w in zip(got, [0.7071, 0.5412, 0.5412, 0.4284, 0.6501]):
        assert g == pytest.approx(w, abs=5e-5)


# ---- random clouds vs the oracle ------------------------------------------------------------------------
WFS = [("uniform", [3.0, 10.0]), ("hyper_exp", [1.0, 0.1]), ("hyper_exp", [0.49, 0.86, 0.55, 0.13, 0.096, 0.157]),
       ("dagum", [1.7, 2.5, 9.0]), ("kumaraswamy", [2.0, 11.0, 3.3, 4.4])]
SDS = [("Hellinger", [2.0]), ("Hellinger", [3.4277149325231795]), ("Kolmogorov-Smirnov", []), ("Kullback-Leibler", [0.514]),
       ("Renyi", [2.428, 0.73]), ("Renyi", [0.0404, 3.566]), ("Renyi", [1.0, 0.5])]


@pytest.mark.parametrize("wf", WFS)
def test_from_primitives_weight_functions(lh, oracle, wf):
    rng = np.random.default_rng(11)
    sa, xa = cloud(rng, 257)
    sb, xb = cloud(rng, 199)
    anchors = [(i, i) for i in range(199)] + [(256, 0), (0, 198), (5, 5)]

    def run(mod):
        lchd = mod.LoCoHD(CATS, mod.WeightFunction(*wf))
        return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, 50.0))

    got, want = both(lh, oracle, run)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < TIGHT


@pytest.mark.parametrize("sd", SDS)
def test_from_primitives_statistical_distances(lh, oracle, sd):
    rng = np.random.default_rng(12)
    sa, xa = cloud(rng, 180)
    sb, xb = cloud(rng, 230)
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, 180, 150), rng.integers(0, 230, 150))]

    def run(mod):
        lchd = mod.LoCoHD(CATS, mod.WeightFunction("hyper_exp", [1.0, 0.08]), statistical_distance=mod.StatisticalDistance(*sd))
        return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, 45.0))

    got, want = both(lh, oracle, run)
    assert np.max(np.abs(got - want)) < 1e-10


def test_tags_weights_and_multi_wf(lh, oracle):
    rng = np.random.default_rng(13)
    n = 240
    sa, xa = cloud(rng, n, box=12.0)
    sb, xb = cloud(rng, n, box=12.0)
    tags = [f"A/{i // 3}-RES" for i in range(n)]
    keys = ["near", "far"]
    anchors = [(i, (7 * i) % n, keys[i % 2]) for i in range(n)]
    for rule in ({"accept_same": False}, {"accept_same": True},
                 {"tag_pairs": {(tags[0], tags[3]), (tags[6], tags[9]), (tags[30], tags[30])}, "accepted_pairs": False, "ordered": True},
                 {"tag_pairs": {(tags[i], tags[j]) for i in range(0, n, 3) for j in range(0, n, 9)}, "accepted_pairs": True, "ordered": False}):

        def run(mod):
            wfd = {"near": mod.WeightFunction("uniform", [3.0, 10.0]), "far": mod.WeightFunction("dagum", [2.0, 5.0, 1.0])}
            lchd = mod.LoCoHD(CATS, wfd, mod.TagPairingRule(rule), category_weights=[1.0, 0.5, 2.25, 3.0, 0.1])
            return np.asarray(lchd.from_primitives(prims(mod, sa, xa, tags), prims(mod, sb, xb, tags), anchors, 9.5))

        got, want = both(lh, oracle, run)
        assert np.max(np.abs(got - want)) < TIGHT, rule


def test_from_coords_and_dmxs(lh, oracle):
    rng = np.random.default_rng(14)
    for n in (1, 2, 63, 64, 65, 100, 700, 1500):
        sa, xa = cloud(rng, n, box=20.0)
        sb, xb = cloud(rng, n, box=20.0)

        def run(mod):
            lchd = mod.LoCoHD(CATS, mod.WeightFunction("hyper_exp", [1.0, 1.0 / 7.0]))
            return np.asarray(lchd.from_coords(sa, sb, xa, xb))

        got, want = both(lh, oracle, run)
        assert np.max(np.abs(got - want)) < TIGHT, n
    # from_dmxs with +inf entries (python_codes/ensembles/compare_ensembles.py:261-263) and unequal widths
    n, m = 90, 70
    sa, xa = cloud(rng, n, box=15.0)
    sb, xb = cloud(rng, m, box=15.0)
    da = np.sqrt(((xa[:60, None, :] - xa[None, :, :]) ** 2).sum(-1))
    db = np.sqrt(((xb[:60, None, :] - xb[None, :, :]) ** 2).sum(-1))
    da[rng.integers(0, 60, 40), rng.integers(0, n, 40)] = np.inf
    for i in range(60):
        da[i, i] = 0.0

    def run2(mod):
        lchd = mod.LoCoHD(CATS, mod.WeightFunction("uniform", [3.0, 10.0]))
        return np.asarray(lchd.from_dmxs(sa, sb, da, db))

    got, want = both(lh, oracle, run2)
    assert np.max(np.abs(got - want)) < TIGHT


def test_ties_and_coincident_points(lh, oracle):
    """Lattice coordinates: many exactly equal distances inside and across the two environments."""
    rng = np.random.default_rng(15)
    grid = np.array(list(itertools.product(range(6), repeat=3)), dtype=float)
    xa = np.concatenate([grid, grid[:20]])  # duplicates => distance-0 neighbours
    xb = grid[rng.permutation(len(grid))]
    sa, sb = rng.choice(CATS, len(xa)).tolist(), rng.choice(CATS, len(xb)).tolist()
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, len(xa), 200), rng.integers(0, len(xb), 200))]
    for wf in (("uniform", [1.0, 3.0]), ("kumaraswamy", [0.0, 4.0, 2.0, 2.0])):

        def run(mod):
            lchd = mod.LoCoHD(CATS, mod.WeightFunction(*wf))
            return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, 3.0000001))

        got, want = both(lh, oracle, run)
        assert np.max(np.abs(got - want)) < TIGHT


def test_self_comparison_is_zero_and_symmetry(lh):
    rng = np.random.default_rng(16)
    s, x = cloud(rng, 400, box=20.0)
    s2, x2 = cloud(rng, 400, box=20.0)
    lchd = lh.LoCoHD(CATS, lh.WeightFunction("hyper_exp", [1.0, 0.2]))
    p, q = prims(lh, s, x), prims(lh, s2, x2)
    anchors = [(i, i) for i in range(400)]
    assert max(abs(v) for v in lchd.from_primitives(p, p, anchors, 12.0)) == 0.0
    ab = np.asarray(lchd.from_primitives(p, q, anchors, 12.0))
    ba = np.asarray(lchd.from_primitives(q, p, anchors, 12.0))
    assert np.max(np.abs(ab - ba)) < 1e-13
    assert ab.min() >= 0.0 and ab.max() <= 1.0 + 1e-12


def test_anchor_order_and_reuse(lh, oracle):
    """Output i must belong to anchor pair i, whatever the order / repetition of anchors."""
    rng = np.random.default_rng(17)
    sa, xa = cloud(rng, 150, box=15.0)
    sb, xb = cloud(rng, 150, box=15.0)
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, 150, 3000), rng.integers(0, 150, 3000))]
    anchors += [(0, 0)] * 5 + [(149, 149), (0, 149), (149, 0)]

    def run(mod):
        lchd = mod.LoCoHD(CATS, mod.WeightFunction("uniform", [3.0, 10.0]))
        return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, 10.0))

    got, want = both(lh, oracle, run)
    assert np.array_equal(np.argsort(got, kind="stable"), np.argsort(want, kind="stable")) or np.max(np.abs(got - want)) < TIGHT
    assert np.max(np.abs(got - want)) < TIGHT


def test_large_environments_retry(lh, oracle):
    """Dense cloud: environments of ~1500 points overflow the default LDS capacity and trigger the retry."""
    rng = np.random.default_rng(18)
    sa, xa = cloud(rng, 3000, box=10.0)
    sb, xb = cloud(rng, 2500, box=10.0)
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, 3000, 64), rng.integers(0, 2500, 64))]

    def run(mod):
        lchd = mod.LoCoHD(CATS, mod.WeightFunction("hyper_exp", [1.0, 0.15]))
        return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, 9.0))

    got, want = both(lh, oracle, run)
    assert np.max(np.abs(got - want)) < TIGHT


def test_many_categories(lh, oracle):
    rng = np.random.default_rng(19)
    for ncat in (1, 8, 9, 16, 17, 25, 32, 33, 64, 65, 200, 255):
        cats = [f"c{i}" for i in range(ncat)]
        sa, xa = cloud(rng, 300, box=12.0, cats=cats)
        sb, xb = cloud(rng, 300, box=12.0, cats=cats)
        anchors = [(i, i) for i in range(300)]

        def run(mod):
            lchd = mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.1]))
            return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, 10.0))

        got, want = both(lh, oracle, run)
        assert np.max(np.abs(got - want)) < TIGHT, ncat


def test_ragged_distance_matrix(lh, oracle):
    """Rows of different lengths: the reference co-sorts each row with a prefix of seq (utils.rs:25-39)."""
    rng = np.random.default_rng(31)
    s, x = cloud(rng, 40, box=8.0)
    full = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    lens = rng.integers(25, 41, 40)
    ragged = []
    for i in range(25):  # row i must contain its own 0 => keep at least i+1 columns
        ragged.append(full[i, : max(int(lens[i]), i + 1)].tolist())
    lchd = lh.LoCoHD(CATS, lh.WeightFunction("hyper_exp", [1.0, 0.3]))
    got = np.asarray(lchd.from_dmxs(s, s, ragged, [r[::1] for r in ragged]))
    assert np.max(np.abs(got)) == 0.0
    lo = oracle.LoCoHD(CATS, oracle.WeightFunction("hyper_exp", [1.0, 0.3]))
    other = full[:25, :40]
    want = []
    for i in range(25):
        (seq_r, d_r), (seq_o, d_o) = _sorted_env(s, ragged[i]), _sorted_env(s, other[i])
        want.append(lo.from_anchors(seq_r, seq_o, d_r, d_o))
    got = np.asarray(lchd.from_dmxs(s, s, ragged, other))
    assert np.max(np.abs(got - np.asarray(want))) < TIGHT


def test_ragged_rows_never_look_beyond_their_length(lh, oracle):
    """What lies beyond a row's length is not part of the environment (utils.rs:25-39 zips the row with seq): categories
    outside the map there do not raise, and statistical distances with logarithms see no padding."""
    rng = np.random.default_rng(32)
    n = 300
    s, x = cloud(rng, n, box=12.0)
    s = list(s)
    for k in range(200, n):
        s[k] = "not-a-category"
    full = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    lens = rng.integers(120, 201, 100)
    ragged_a = [full[i, : int(lens[i])].tolist() for i in range(100)]
    ragged_b = [full[(i + 7) % 100, : int(lens[(i * 3) % 100])].tolist() for i in range(100)]
    for r in ragged_b:
        r[0], r[int(np.argmin(r))] = 0.0, r[0]  # the anchor (distance 0) first, as stat_dist_integral demands
    for sd in (("Hellinger", [2.0]), ("Kullback-Leibler", [1e-3]), ("Renyi", [2.0, 1e-3])):
        lchd = lh.LoCoHD(CATS, lh.WeightFunction("hyper_exp", [1.0, 0.3]), statistical_distance=lh.StatisticalDistance(*sd))
        lo = oracle.LoCoHD(CATS, oracle.WeightFunction("hyper_exp", [1.0, 0.3]), statistical_distance=oracle.StatisticalDistance(*sd))
        got = np.asarray(lchd.from_dmxs(s, s, ragged_a, ragged_b))
        want = []
        for i in range(100):
            (sa, da), (sb, db) = _sorted_env(s, ragged_a[i]), _sorted_env(s, ragged_b[i])
            want.append(lo.from_anchors(sa, sb, da, db))
        assert np.all(np.isfinite(got)), sd
        assert np.max(np.abs(got - np.asarray(want))) < TIGHT, sd
    # a row that DOES reach an unknown category raises like the reference (pmf.rs:38-42)
    ragged_a[5] = full[5, :201].tolist()
    with pytest.raises(ValueError):
        lchd.from_dmxs(s, s, ragged_a, ragged_b)


def _sorted_env(seq, row):
    row = np.asarray(row, dtype=float)
    order = np.argsort(row, kind="stable")
    return [seq[k] for k in order], row[order].tolist()


def test_big_environments_block_kernel(lh, oracle):
    """Environments of ~6000 points: the multi-wave environment kernel (capacity 8192) and the global sqrt tables."""
    rng = np.random.default_rng(37)
    sa, xa = cloud(rng, 7000, box=6.0)
    sb, xb = cloud(rng, 6500, box=6.0)
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, 7000, 6), rng.integers(0, 6500, 6))]

    def run(mod):
        lchd = mod.LoCoHD(CATS, mod.WeightFunction("hyper_exp", [1.0, 0.2]))
        return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, 11.0))

    got, want = both(lh, oracle, run)
    assert np.max(np.abs(got - want)) < TIGHT


def test_environments_beyond_16384_points(lh, oracle):
    """A threshold that swallows most of a 20 000-atom structure: environments of ~8 000 .. 20 000 points.  Up to 16 384 they
    are sorted in LDS by the 1024-thread environment kernel; beyond, k_env_collect appends them unsorted to scratch rows and
    the global-memory row sort finishes them (capacity 32 768).  With a tag rule and a dictionary of weight functions."""
    rng = np.random.default_rng(61)
    n = 20_000
    sa, xa = cloud(rng, n, box=20.0)
    sb, xb = cloud(rng, n, box=20.0)
    tags = [f"r{i // 5}" for i in range(n)]
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, n, 5), rng.integers(0, n, 5))]
    for thr, rule, multi in ((float("inf"), None, False), (30.0, {"accept_same": False}, True), (18.0, None, False)):

        def run(mod):
            wf = {"a": mod.WeightFunction("hyper_exp", [1.0, 0.2]), "b": mod.WeightFunction("uniform", [1.0, 25.0])} if multi \
                else mod.WeightFunction("hyper_exp", [1.0, 0.2])
            lchd = mod.LoCoHD(CATS, wf, None if rule is None else mod.TagPairingRule(rule))
            ap = [(i, j, "a" if k % 2 else "b") for k, (i, j) in enumerate(anchors)] if multi else anchors
            return np.asarray(lchd.from_primitives(prims(mod, sa, xa, tags), prims(mod, sb, xb, tags), ap, thr))

        got, want = both(lh, oracle, run)
        assert np.max(np.abs(got - want)) < TIGHT, thr


def test_two_contexts_and_capacity_decay(lh, oracle):
    """Two contexts alive at once: the dynamic-LDS attributes and the hooks belong to a context / its device, not to the
    process, so a > 4096-point environment (the 1024-thread environment kernel with > 64 KB of LDS) must work on the SECOND
    context as well; a context whose capacity hint was raised by one dense call keeps giving right answers while the hint
    decays again over the following small calls."""
    rng = np.random.default_rng(53)
    sa, xa = cloud(rng, 7000, box=6.0)
    sb, xb = cloud(rng, 6500, box=6.0)
    big_anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, 7000, 4), rng.integers(0, 6500, 4))]
    ss, xs = cloud(rng, 400, box=12.0)
    small_anchors = [(i, i) for i in range(0, 400, 3)]

    def big(mod, lchd):
        return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), big_anchors, 11.0))

    def small(mod, lchd):
        return np.asarray(lchd.from_primitives(prims(mod, ss, xs), prims(mod, ss, xs[::-1].copy()), small_anchors, 9.0))

    mk = lambda mod: mod.LoCoHD(CATS, mod.WeightFunction("hyper_exp", [1.0, 0.2]))
    first, second = mk(lh), mk(lh)
    ref = mk(oracle)
    want_big, want_small = big(oracle, ref), small(oracle, ref)
    assert np.max(np.abs(small(lh, first) - want_small)) < TIGHT   # creates context 1
    assert np.max(np.abs(big(lh, second) - want_big)) < TIGHT      # context 2: capacity 8192 on its first call
    assert np.max(np.abs(big(lh, first) - want_big)) < TIGHT
    for _ in range(40):  # the raised capacity hint halves every eight small passes: 8192 -> ... -> 512
        assert np.max(np.abs(small(lh, second) - want_small)) < TIGHT
    assert np.max(np.abs(big(lh, second) - want_big)) < TIGHT


def test_three_hundred_categories(lh, oracle):
    """The reference's category map is an arbitrary HashMap (/root/reference/src/locohd.rs:312-316).  Beyond 255 categories
    the ids travel as 16 bits (k_env_cells<.., uint16_t>, k_sweep_wide<.., CAT16>): from_anchors and from_primitives, every
    statistical-distance family, category weights, a weight-function dictionary, both tag rules; from_coords / from_dmxs in
    test_three_hundred_categories_dense below; more than 512: test_more_than_512_categories."""
    rng = np.random.default_rng(61)
    cats = [f"t{i}" for i in range(300)]
    n = 500
    sa, xa = rng.choice(cats, n).tolist(), rng.uniform(0, 22.0, (n, 3))
    sb, xb = rng.choice(cats, n).tolist(), rng.uniform(0, 22.0, (n, 3))
    sa[0], sb[0], sa[1], sb[1] = "t299", "t299", "t255", "t256"  # ids on both sides of the one-byte boundary among the anchors
    tags = [f"r{i // 5}" for i in range(n)]
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, n, 300), rng.integers(0, n, 300))] + [(0, 0), (1, 1), (0, 1)]
    w = rng.uniform(0.5, 2.0, 300).tolist()
    cases = [dict(), dict(category_weights=w), dict(rule={"accept_same": False}),
             dict(rule={"tag_pairs": {("r1", "r2"), ("r3", "r3"), ("r10", "r40")}, "accepted_pairs": False, "ordered": False}),
             dict(sd=("Kolmogorov-Smirnov", [])), dict(sd=("Kullback-Leibler", [1e-9])), dict(sd=("Renyi", [1.7, 1e-9])), dict(sd=("Hellinger", [3.0])),
             dict(sd=("Hellinger", [2.0]), category_weights=w)]
    for case in cases:

        def run(mod):
            kw = {}
            if "category_weights" in case:
                kw["category_weights"] = case["category_weights"]
            if "sd" in case:
                kw["statistical_distance"] = mod.StatisticalDistance(*case["sd"])
            rule = mod.TagPairingRule(case["rule"]) if "rule" in case else None
            lchd = mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.15]), rule, **kw)
            out = [np.asarray(lchd.from_primitives(prims(mod, sa, xa, tags), prims(mod, sb, xb, tags), anchors, 8.0)) for _ in range(2)]
            da, db = np.sort(rng_d.uniform(0, 9, 40)), np.sort(rng_d.uniform(0, 9, 55))
            da[0] = db[0] = 0.0
            one = lchd.from_anchors(sa[:40], sb[:55], da.tolist(), db.tolist())
            return out, one

        rng_d = np.random.default_rng(7)
        got, got1 = run(lh)
        rng_d = np.random.default_rng(7)
        want, want1 = run(oracle)
        for g in got:
            assert np.max(np.abs(g - want[0])) < TIGHT, case
        assert abs(got1 - want1) < TIGHT, case
    # a weight-function dictionary (the sweep evaluates the CDFs itself)
    def run_multi(mod):
        lchd = mod.LoCoHD(cats, {"a": mod.WeightFunction("uniform", [2.0, 8.0]), "b": mod.WeightFunction("kumaraswamy", [1.0, 9.0, 2.0, 3.0])})
        keyed = [(i, j, "a" if (i + j) % 2 else "b") for i, j in anchors]
        return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), keyed, 8.0))
    assert np.max(np.abs(run_multi(lh) - run_multi(oracle))) < TIGHT
    # a label outside the map is still an error; the remaining refusals are loud
    lchd = lh.LoCoHD(cats, lh.WeightFunction("uniform", [0.0, 4.0]))
    with pytest.raises(ValueError):
        lchd.from_primitives(prims(lh, ["t1", "nope"], xa[:2] * 0.01), prims(lh, ["t1", "t2"], xb[:2] * 0.01), [(0, 0)], 8.0)
    with pytest.raises(NotImplementedError):  # (ids travel as 16 bits, 0xFFFF = not in the map)
        lh.LoCoHD([f"c{i}" for i in range(65535)]).from_anchors(["c0"], ["c0"], [0.0], [0.0])


@pytest.mark.parametrize("n_cat", [513, 1500, 5000])
def test_more_than_512_categories(lh, oracle, n_cat):
    """Beyond 512 categories the sweep's per-category state no longer fits LDS: k_sweep_wide<.., HUGE> keeps the per-lane count
    columns, the carry row and the generic distances' normalised vectors in a global-memory block per workgroup (the reference's
    category map has no size limit, src/locohd.rs:312-316).  from_primitives, from_anchors, from_coords and from_dmxs; every distance
    family, category weights, a weight-function dictionary, a tag rule."""
    rng = np.random.default_rng(n_cat)
    cats = [f"t{i}" for i in range(n_cat)]
    n = 400
    sa, xa = rng.choice(cats, n).tolist(), rng.uniform(0, 20.0, (n, 3))
    sb, xb = rng.choice(cats, n).tolist(), rng.uniform(0, 20.0, (n, 3))
    sa[0], sb[0], sa[1], sb[1] = cats[-1], cats[-1], "t511", "t512"
    tags = [f"r{i // 5}" for i in range(n)]
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, n, 120), rng.integers(0, n, 120))] + [(0, 0), (1, 1), (0, 1)]
    w = rng.uniform(0.5, 2.0, n_cat).tolist()
    cases = [dict(), dict(category_weights=w), dict(rule={"accept_same": False}), dict(sd=("Kolmogorov-Smirnov", [])),
             dict(sd=("Kullback-Leibler", [1e-3])), dict(sd=("Renyi", [1.7, 1e-9])), dict(sd=("Hellinger", [3.0]), category_weights=w)]
    if n_cat > 2000:
        cases = cases[:2] + cases[3:5]
    for case in cases:

        def run(mod):
            kw = {}
            if "category_weights" in case:
                kw["category_weights"] = case["category_weights"]
            if "sd" in case:
                kw["statistical_distance"] = mod.StatisticalDistance(*case["sd"])
            rule = mod.TagPairingRule(case["rule"]) if "rule" in case else None
            lchd = mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.15]), rule, **kw)
            out = [np.asarray(lchd.from_primitives(prims(mod, sa, xa, tags), prims(mod, sb, xb, tags), anchors, 8.0)) for _ in range(2)]
            da, db = np.sort(rng_d.uniform(0, 9, 40)), np.sort(rng_d.uniform(0, 9, 55))
            da[0] = db[0] = 0.0
            one = lchd.from_anchors(sa[:40], sb[:55], da.tolist(), db.tolist())
            dense = np.asarray(lchd.from_coords(sa[:150], sb[:150], xa[:150], xb[:150]))
            return out, one, dense

        rng_d = np.random.default_rng(7)
        got, got1, gotd = run(lh)
        rng_d = np.random.default_rng(7)
        want, want1, wantd = run(oracle)
        for g in got:
            assert np.max(np.abs(g - want[0])) < TIGHT, case
        assert abs(got1 - want1) < TIGHT, case
        assert np.max(np.abs(gotd - wantd)) < TIGHT, case

    def run_multi(mod):
        lchd = mod.LoCoHD(cats, {"a": mod.WeightFunction("uniform", [2.0, 8.0]), "b": mod.WeightFunction("kumaraswamy", [1.0, 9.0, 2.0, 3.0])})
        keyed = [(i, j, "a" if (i + j) % 2 else "b") for i, j in anchors]
        return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), keyed, 8.0))
    assert np.max(np.abs(run_multi(lh) - run_multi(oracle))) < TIGHT
    lchd = lh.LoCoHD(cats, lh.WeightFunction("uniform", [0.0, 4.0]))
    with pytest.raises(ValueError):  # a label outside the map is still an error
        lchd.from_primitives(prims(lh, ["t1", "nope"], xa[:2] * 0.01), prims(lh, ["t1", "t2"], xb[:2] * 0.01), [(0, 0)], 8.0)


@pytest.mark.parametrize("n", [1, 7, 300, 1100, 9000])
def test_three_hundred_categories_dense(lh, oracle, n):
    """from_coords / from_dmxs (square and ragged) with 300 categories: k_env_rows<.., uint16_t> (rows of up to 8192 points sorted
    in LDS, longer ones in the environment store) + k_sweep_wide<.., CAT16>.  Against the oracle (sampled rows at 9000)."""
    rng = np.random.default_rng(300 + n)
    cats = [f"t{i}" for i in range(300)]
    sa, sb = rng.choice(cats, n).tolist(), rng.choice(cats, n).tolist()
    if n > 2:
        sa[0], sb[0], sa[1], sb[1] = "t299", "t254", "t255", "t256"  # ids on both sides of the one-byte boundary
    side = (n / 0.05) ** (1 / 3) + 1.0
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    wf = ("hyper_exp", [1.0, 0.2])
    g, o = lh.LoCoHD(cats, lh.WeightFunction(*wf)), oracle.LoCoHD(cats, oracle.WeightFunction(*wf), n_of_threads=8)
    got = np.asarray(g.from_coords(sa, sb, xa, xb))
    if n <= 1100:
        want = np.asarray(o.from_coords(sa, sb, xa, xb))
        assert np.max(np.abs(got - want)) < TIGHT
    else:  # the oracle on a sample of rows: from_anchors on the row's own sorted distances
        for r in rng.integers(0, n, 12):
            da, db = np.linalg.norm(xa - xa[r], axis=1), np.linalg.norm(xb - xb[r], axis=1)
            ia, ib = np.argsort(da, kind="stable"), np.argsort(db, kind="stable")
            want = o.from_anchors([sa[i] for i in ia], [sb[i] for i in ib], da[ia].tolist(), db[ib].tolist())
            assert abs(got[r] - want) < 1e-11, r
    if n > 1100:
        return
    # given distance matrices, other distance families, category weights
    dma = np.linalg.norm(xa[:, None] - xa[None], axis=2)
    dmb = np.linalg.norm(xb[:, None] - xb[None], axis=2)
    w = rng.uniform(0.5, 2.0, 300).tolist()
    for kw in (dict(), dict(category_weights=w), dict(sd=("Kolmogorov-Smirnov", [])), dict(sd=("Renyi", [1.7, 1e-9]))):
        def make(mod):
            k = {key: val for key, val in kw.items() if key != "sd"}
            if "sd" in kw:
                k["statistical_distance"] = mod.StatisticalDistance(*kw["sd"])
            return mod.LoCoHD(cats, mod.WeightFunction(*wf), **k)
        got = np.asarray(make(lh).from_dmxs(sa, sb, dma.tolist(), dmb.tolist()))
        want = np.asarray(make(oracle).from_dmxs(sa, sb, dma.tolist(), dmb.tolist()))
        assert np.max(np.abs(got - want)) < TIGHT, kw
    if n >= 7:  # ragged rows: row r is sorted with a prefix of seq (utils.rs:25-39)
        ra = [sorted(rng.uniform(0, 9, int(k)).tolist()) for k in rng.integers(1, n + 1, n)]
        rb = [sorted(rng.uniform(0, 9, int(k)).tolist()) for k in rng.integers(1, n + 1, n)]
        for row in ra + rb:
            row[0] = 0.0
        got = np.asarray(g.from_dmxs(sa, sb, ra, rb))
        want = np.asarray([o.from_anchors(sa[: len(x)], sb[: len(y)], x, y) for x, y in zip(ra, rb)])  # (the rows are ascending already)
        assert np.max(np.abs(got - want)) < TIGHT


@pytest.mark.parametrize("ncat", [5, 40, 200])
def test_from_anchors_beyond_65535_points(lh, oracle, ncat):
    """The reference sweeps lists of any length (/root/reference/src/locohd.rs:61-226).  Beyond 65 535 points the counts of a category
    no longer fit the 16-bit fields of the regular sweeps: the 64-bit-count form of the wide sweep (k_sweep_wide<.., BIG>) takes over,
    with square roots beyond the 65 536-entry tables computed.  One dominant category makes a single count pass 65 535."""
    rng = np.random.default_rng(65 + ncat)
    cats = [f"c{i}" for i in range(ncat)]
    w = rng.uniform(0.5, 2.0, ncat).tolist()
    for na, nb in ((70_000, 300), (66_000, 131_500), (65_536, 65_535)):
        pa = np.full(ncat, 0.1 / max(ncat - 1, 1)); pa[0] = 0.9  # ~90 % of side A in one category
        sa = [cats[i] for i in rng.choice(ncat, na, p=pa / pa.sum())]
        sb = [cats[i] for i in rng.integers(0, ncat, nb)]
        da, db = np.sort(rng.uniform(0, 30, na)), np.sort(rng.uniform(0, 30, nb))
        da[0] = db[0] = 0.0
        da[5:9] = da[5]  # ties inside a list and across the lists
        db[3] = da[5]
        db = np.sort(db)
        for kw in (dict(), dict(category_weights=w), dict(sd=("Kolmogorov-Smirnov", [])), dict(sd=("Hellinger", [3.0]))):
            def make(mod):
                k = {key: val for key, val in kw.items() if key != "sd"}
                if "sd" in kw:
                    k["statistical_distance"] = mod.StatisticalDistance(*kw["sd"])
                return mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.2]), **k)
            got = make(lh).from_anchors(sa, sb, da.tolist(), db.tolist())
            want = make(oracle).from_anchors(sa, sb, da.tolist(), db.tolist())
            assert abs(got - want) < 1e-11, (na, nb, kw)
        if ncat == 5:  # a weight-function dictionary (the sweep evaluates the CDF itself)
            mk = lambda mod: mod.LoCoHD(cats, {"u": mod.WeightFunction("uniform", [3.0, 20.0]), "k": mod.WeightFunction("kumaraswamy", [1.0, 28.0, 2.0, 3.0])})
            assert abs(mk(lh).from_anchors(sa, sb, da.tolist(), db.tolist(), "k") - mk(oracle).from_anchors(sa, sb, da.tolist(), db.tolist(), "k")) < 1e-11
    with pytest.raises(NotImplementedError):  # more than 255 categories AND more than 65 535 points: still refused, loudly
        many = [f"c{i}" for i in range(300)]
        lh.LoCoHD(many).from_anchors(["c0"] * 70_000, ["c1"], [0.0] * 70_000, [0.0])


def test_from_dmxs_rows_beyond_65535_points(lh, oracle):
    """Distance rows of more than 65 535 entries (the reference co-sorts any length, utils.rs:25-39): sorted in the environment store
    by k_env_rows<1024, GLOBALKV> with 32 768 buckets, swept by k_sweep_wide<.., BIG>.  Square and ragged, three rows each (from_coords
    takes the same path, but its n x n rows at this size need > 100 GB of store: not a test)."""
    rng = np.random.default_rng(6553)
    cats = [f"c{i}" for i in range(12)]
    na, nb = 70_000, 90_000
    pa = np.full(12, 0.01); pa[3] = 0.89
    sa = [cats[i] for i in rng.choice(12, na, p=pa / pa.sum())]
    sb = [cats[i] for i in rng.integers(0, 12, nb)]
    rows = 3
    ma = rng.gamma(3.0, 4.0, (rows, na))  # (a skewed distance distribution: uneven buckets)
    mb = rng.uniform(0, 40, (rows, nb))
    ma[:, 7] = 0.0
    mb[:, 11] = 0.0
    ma[1, 100:4100] = ma[1, 100]  # thousands of equal distances in one row: the fuller-bucket path
    g = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1]))
    o = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.1]))
    got = np.asarray(g.from_dmxs(sa, sb, ma, mb))
    for r in range(rows):
        ia, ib = np.argsort(ma[r], kind="stable"), np.argsort(mb[r], kind="stable")
        want = o.from_anchors([sa[i] for i in ia], [sb[i] for i in ib], ma[r][ia].tolist(), mb[r][ib].tolist())
        assert abs(got[r] - want) < 1e-11, r
    # ragged: row r is sorted with a prefix of seq
    lens_a, lens_b = [na, 66_000, 10], [70_001, nb, 65_536]
    ra = [ma[r, :k].tolist() for r, k in enumerate(lens_a)]
    rb = [mb[r, :k].tolist() for r, k in enumerate(lens_b)]
    ra[2][0] = 0.0
    got = np.asarray(g.from_dmxs(sa, sb, ra, rb))
    for r in range(rows):
        x, y = np.asarray(ra[r]), np.asarray(rb[r])
        ia, ib = np.argsort(x, kind="stable"), np.argsort(y, kind="stable")
        want = o.from_anchors([sa[i] for i in ia], [sb[i] for i in ib], x[ia].tolist(), y[ib].tolist())
        assert abs(got[r] - want) < 1e-11, r


def test_from_primitives_environment_beyond_65535_points(lh, oracle):
    """A threshold that swallows a 70 000-atom cloud: environments of more than 65 535 points are collected unsorted (k_env_collect),
    sorted in the store (k_env_rows<1024, GLOBALKV>, 32 768 buckets) and swept with 64-bit count words (k_sweep_wide<.., BIG>);
    the capacity grows 512 -> ... -> 131 072 through the overflow retries of the pass."""
    rng = np.random.default_rng(70_000)
    n = 70_000
    cats = [f"c{i}" for i in range(6)]
    v = rng.normal(size=(n, 3))
    xa = v / np.linalg.norm(v, axis=1)[:, None] * (18.0 * rng.uniform(0, 1, n)[:, None] ** (1 / 3))
    v = rng.normal(size=(n, 3))
    xb = v / np.linalg.norm(v, axis=1)[:, None] * (18.0 * rng.uniform(0, 1, n)[:, None] ** (1 / 3))
    ca, cb = rng.choice(6, n, p=[0.95, 0.01, 0.01, 0.01, 0.01, 0.01]).astype(np.int32), rng.integers(0, 6, n).astype(np.int32)
    pairs = np.array([[0, 5], [17, 17], [n - 1, 3], [0, 3]], dtype=np.int64)
    for rule, tag in ((None, np.zeros(n, dtype=np.int32)), ({"accept_same": False}, (np.arange(n) // 7).astype(np.int32))):
        lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.1]), oracle.TagPairingRule(rule) if rule else None)
        lg = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1]), lh.TagPairingRule(rule) if rule else None)
        want, sizes = lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, 40.0, return_env_sizes=True)
        assert np.min(sizes) > 65_535
        pk = lambda x, c: lh.api._Packed(x, c, tag)
        got = lg.from_packed(pk(xa, ca), pk(xb, cb), pairs, 40.0)
        assert np.max(np.abs(np.asarray(got) - np.asarray(want))) < 1e-11, rule


def test_error_behaviour(lh):
    lchd = lh.LoCoHD(["A", "B"], lh.WeightFunction("uniform", [0.0, 4.0]))
    with pytest.raises(ValueError):  # src/locohd.rs:70-73
        lchd.from_anchors(["A", "B"], ["A"], [0.0], [0.0])
    with pytest.raises(ValueError):  # :74-77
        lchd.from_anchors(["A", "B"], ["A"], [0.5, 1.0], [0.0])
    with pytest.raises(ValueError):  # pmf.rs:38-42 unknown category
        lchd.from_anchors(["A", "Z"], ["A"], [0.0, 1.0], [0.0])
    with pytest.raises(ValueError):  # :276-281 key given for a single weight function
        lchd.from_anchors(["A"], ["A"], [0.0], [0.0], "k")
    with pytest.raises(ValueError):  # empty anchor list is the 3-tuple variant (:34-40)
        lchd.from_primitives([], [], [], 5.0)
    p = [lh.PrimitiveAtom("A", "", [0.0, 0.0, 0.0]), lh.PrimitiveAtom("Z", "", [1.0, 0.0, 0.0])]
    with pytest.raises(ValueError):  # unknown category inside an environment
        lchd.from_primitives(p, p, [(0, 0)], 5.0)
    assert lchd.from_primitives(p, p, [(0, 0)], 0.5) == [0.0]  # ...but not when it is outside every environment
    with pytest.raises(lh.PanicException):  # anchor out of range (:521)
        lchd.from_primitives(p, p, [(0, 2)], 5.0)
    with pytest.raises(lh.PanicException):  # non-positive threshold => empty environments (:74)
        lchd.from_primitives(p, p, [(0, 0)], 0.0)
    with pytest.raises(ValueError):  # :420-428
        lchd.from_dmxs(["A"], ["A"], [[0.0]], [[0.0], [0.0]])
    with pytest.raises(ValueError):  # row without a zero: dists[0] != 0 (collapsed message, :448-452)
        lchd.from_dmxs(["A", "B"], ["A", "B"], [[1.0, 2.0]], [[0.0, 1.0]])
    multi = lh.LoCoHD(["A", "B"], {"x": lh.WeightFunction("uniform", [0.0, 4.0])})
    assert multi.from_primitives(p[:1], p[:1], [], 5.0) == []
    with pytest.raises(ValueError):
        multi.from_primitives(p[:1], p[:1], [(0, 0, "nope")], 5.0)
    assert multi.from_primitives(p[:1], p[:1], [(0, 0, "x")], 5.0) == [0.0]


@pytest.mark.parametrize("hook", [{"LCHD_FORCE_CMAX": "8"}, {"LCHD_FORCE_CMAX": "12"}, {"LCHD_FORCE_CMAX": "16"},
                                  {"LCHD_FORCE_CMAX": "20"}, {"LCHD_FORCE_CMAX": "24"}, {"LCHD_FORCE_CMAX": "28"}, {"LCHD_FORCE_CMAX": "32"}, {"LCHD_FORCE_GENERIC": "1"},
                                  {"LCHD_FORCE_BIGENV": "1"}, {"LCHD_FORCE_GENERIC": "1", "LCHD_FORCE_CMAX": "16"},
                                  {"LCHD_FORCE_WIDE": "1"}, {"LCHD_FORCE_WIDE": "1", "LCHD_FORCE_GENERIC": "1"},
                                  {"LCHD_NO_CDF_KEYS": "1"}, {"LCHD_NO_CDF_KEYS": "1", "LCHD_FORCE_WIDE": "1"}])
def test_every_sweep_kernel_variant(lh, oracle, hook, monkeypatch):
    """The sweep kernel is instantiated per category-slot count / distance family / table placement; the launcher's
    test hooks force each instantiation onto the same inputs (an -O3 miscompile of one variant was caught this way)."""
    rng = np.random.default_rng(23)
    sa, xa = cloud(rng, 500, box=13.0)
    sb, xb = cloud(rng, 450, box=13.0)
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, 500, 400), rng.integers(0, 450, 400))]
    for k, v in hook.items():
        monkeypatch.setenv(k, v)
    for wf, w in ((("hyper_exp", [1.0, 0.1]), None), (("kumaraswamy", [2.0, 11.0, 3.3, 4.4]), [1.0, 0.5, 2.25, 3.0, 0.1])):

        def run(mod):
            lchd = mod.LoCoHD(CATS, mod.WeightFunction(*wf), category_weights=w)
            return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, 9.0))

        want = run(oracle)
        # a call this small takes the one-launch sweep (records worked out inline) where the configuration allows it; the
        # second round forces the regular pipeline (k_pair_meta + k_sweep / k_sweep_duo) onto the same inputs, the third the
        # regular pipeline without the two-pairs-per-wavefront kernel
        for extra in ({}, {"LCHD_NO_INLINE_META": "1"}, {"LCHD_NO_INLINE_META": "1", "LCHD_NO_DUO": "1"}):
            for k, v in extra.items():
                monkeypatch.setenv(k, v)
            got = run(lh)
            for k in extra:
                monkeypatch.delenv(k)
            assert np.max(np.abs(got - want)) < TIGHT, (hook, extra, wf)


def test_near_identical_environments(lh, oracle):
    """Small Hellinger distances exercise the exact difference-of-roots branch of the fast path."""
    rng = np.random.default_rng(29)
    s, x = cloud(rng, 600, box=14.0)
    x2 = x + rng.normal(0.0, 0.02, x.shape)  # slightly jittered copy, same categories
    s3 = list(s)
    for i in rng.integers(0, 600, 12):
        s3[i] = CATS[(CATS.index(s3[i]) + 1) % len(CATS)]  # a few relabelled atoms
    anchors = [(i, i) for i in range(600)]
    for sb_, xb_ in ((s, x2), (s3, x), (s3, x2)):

        def run(mod):
            lchd = mod.LoCoHD(CATS, mod.WeightFunction("uniform", [3.0, 10.0]))
            return np.asarray(lchd.from_primitives(prims(mod, s, x), prims(mod, sb_, xb_), anchors, 10.0))

        got, want = both(lh, oracle, run)
        assert np.max(np.abs(got - want)) < TIGHT


def test_degenerate_geometries(lh, oracle):
    """Inputs that defeat the balanced-bucket assumptions of the sorting kernels (they must only get slower)."""
    rng = np.random.default_rng(41)
    cases = {
        "all_identical": (np.zeros((300, 3)), np.zeros((280, 3)) + 0.5),
        "two_clusters": (np.concatenate([rng.normal(0, 0.01, (200, 3)), rng.normal(8, 0.01, (200, 3))]),
                         np.concatenate([rng.normal(0, 0.01, (150, 3)), rng.normal(8, 0.01, (250, 3))])),
        "collinear": (np.stack([np.linspace(0, 30, 350), np.zeros(350), np.zeros(350)], 1),
                      np.stack([np.linspace(0, 30, 350) ** 1.1, np.zeros(350), np.zeros(350)], 1)),
        "far_from_origin": (rng.uniform(0, 20, (300, 3)) + 1.0e6, rng.uniform(0, 20, (300, 3)) - 1.0e6),
        "single_atoms": (np.array([[1.0, 2.0, 3.0]]), np.array([[0.0, 0.0, 0.0]])),
    }
    for name, (xa, xb) in cases.items():
        sa, sb = rng.choice(CATS, len(xa)).tolist(), rng.choice(CATS, len(xb)).tolist()
        anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, len(xa), 120), rng.integers(0, len(xb), 120))]
        for thr in (9.0, float("inf")):

            def run(mod):
                lchd = mod.LoCoHD(CATS, mod.WeightFunction("hyper_exp", [1.0, 0.2]))
                return np.asarray(lchd.from_primitives(prims(mod, sa, xa), prims(mod, sb, xb), anchors, thr))

            got, want = both(lh, oracle, run)
            assert np.max(np.abs(got - want)) < TIGHT, (name, thr)
        if len(xa) == len(xb):

            def run_c(mod):
                lchd = mod.LoCoHD(CATS, mod.WeightFunction("uniform", [0.5, 12.0]))
                return np.asarray(lchd.from_coords(sa, sb, xa, xb))

            got, want = both(lh, oracle, run_c)
            assert np.max(np.abs(got - want)) < TIGHT, name


def test_points_exactly_on_the_threshold(lh, oracle):
    """Lattice points at distance exactly == threshold are outside (strict d^2 < thr^2, kd-tree 0.6 within_radius);
    both paths must agree on every such boundary case."""
    grid = np.array(list(itertools.product(range(-4, 5), repeat=3)), dtype=float)
    rng = np.random.default_rng(43)
    s = rng.choice(CATS, len(grid)).tolist()
    s2 = rng.choice(CATS, len(grid)).tolist()
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, len(grid), 150), rng.integers(0, len(grid), 150))]
    for thr in (1.0, 2.0, 3.0, 5.0 ** 0.5, 3.0000000000000004):

        def run(mod):
            lchd = mod.LoCoHD(CATS, mod.WeightFunction("uniform", [0.0, 4.0]))
            return np.asarray(lchd.from_primitives(prims(mod, s, grid), prims(mod, s2, grid), anchors, thr))

        got, want = both(lh, oracle, run)
        assert np.max(np.abs(got - want)) < TIGHT, thr


def test_from_anchors_on_lists_that_do_not_ascend(lh, oracle):
    """The reference checks only dists[0] (src/locohd.rs:70-77) and its two-pointer loop still computes a well-defined number on
    lists whose distances do not ascend (heads compared as they come, the tail walked in list order, the first tail interval
    starting at the LAST element of the finished list, :134-221).  One lane walks the same loop on the device."""
    rng = np.random.default_rng(123)
    wfs = [("uniform", [0.0, 6.0]), ("hyper_exp", [1.0, 0.3]), ("kumaraswamy", [1.0, 9.0, 2.0, 3.0]), ("dagum", [2.0, 4.0, 1.5])]
    sds = [None, ("Kolmogorov-Smirnov", []), ("Kullback-Leibler", [1e-3]), ("Renyi", [2.5, 1e-3]), ("Hellinger", [3.0])]
    for trial in range(24):
        na, nb = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        da = np.concatenate([[0.0], rng.uniform(0.0, 8.0, na - 1)])  # NOT sorted
        db = np.concatenate([[0.0], rng.uniform(0.0, 8.0, nb - 1)])
        if trial % 4 == 0 and na > 3 and nb > 3:
            db[1:3] = da[1:3]  # equal heads: the branch that advances both lists
        sa, sb = [CATS[k] for k in rng.integers(0, 5, na)], [CATS[k] for k in rng.integers(0, 5, nb)]
        wf, sd = wfs[trial % 4], sds[trial % 5]
        weights = None if trial % 3 else [1.0, 0.5, 2.25, 3.0, 0.1]

        def run(mod):
            kw = {}
            if sd is not None:
                kw["statistical_distance"] = mod.StatisticalDistance(*sd)
            if weights is not None:
                kw["category_weights"] = weights
            return mod.LoCoHD(CATS, mod.WeightFunction(*wf), **kw).from_anchors(sa, sb, da.tolist(), db.tolist())

        got, want = run(lh), run(oracle)
        assert abs(got - want) < TIGHT, (trial, na, nb, wf, sd)
    # an ascending pair of lists still takes the sweep kernel and agrees with the literal walk of the oracle
    da, db = np.sort(da), np.sort(db)
    assert abs(lh.LoCoHD(CATS, lh.WeightFunction("uniform", [0.0, 6.0])).from_anchors(sa, sb, da.tolist(), db.tolist())
               - oracle.LoCoHD(CATS, oracle.WeightFunction("uniform", [0.0, 6.0])).from_anchors(sa, sb, da.tolist(), db.tolist())) < TIGHT
