"""Side-B environments that are used ONCE (the frames of a trajectory, (i, i) lists, a rank's partners under strong scaling).  The
reference builds and consumes an environment inside one closure per pair and side (/root/reference/src/locohd.rs:514-554); here

the pipeline skips the de-duplication of such a side: environment slot p belongs to pair p (picked by the library when the previous
regular pass found (almost) every side-B anchor unique; LCHD_PER_PAIR=1 / -1 force it on / off).  Forced here onto inputs the CPU
oracle can follow.  (Round 5's k_env_sweep -- environment build + sweep in one kernel -- measured 12 % slower than the two kernels
and left the tree in round 6)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT = 1e-11


@pytest.fixture(scope="module")
def lh():
    import loco_hd_amd

    return loco_hd_amd


def _cloud(rng, n, side, n_cat, n_tags=0):
    xyz = rng.uniform(0.0, side, (n, 3))
    cat = rng.integers(0, n_cat, n).astype(np.int32)
    tag = (np.arange(n) // 3).astype(np.int32) if n_tags == 0 else rng.integers(0, n_tags, n).astype(np.int32)
    return xyz, cat, tag


def _run(lh, monkeypatch, build, xa, ca, ta, xb, cb, tb, pairs, thr, repeat=2, per_pair=None):
    import torch
    from loco_hd_amd.device import DeviceSession

    if per_pair is None:
        monkeypatch.delenv("LCHD_PER_PAIR", raising=False)
    else:
        monkeypatch.setenv("LCHD_PER_PAIR", per_pair)
    sess = DeviceSession(build(lh))  # (the hooks are read when the context is created)
    monkeypatch.delenv("LCHD_PER_PAIR", raising=False)
    a, b = sess.upload(xa, ca, ta), sess.upload(xb, cb, tb)
    anchors = torch.from_numpy(np.ascontiguousarray(pairs, dtype=np.int64)).cuda()
    outs = []
    for _ in range(repeat):
        out = torch.full((len(pairs),), -3.0, dtype=torch.float64, device="cuda")
        outs.append(sess.from_primitives(a, b, anchors, thr, out=out).cpu().numpy())
    counts = sess.pass_counts()
    sess.close()
    return outs, counts


@pytest.mark.parametrize("n_cat", [5, 12, 16, 20, 28])
def test_forced_per_pair_side_b_matches_the_oracle_and_the_regular_pipeline(lh, oracle, monkeypatch, n_cat):
    """(i, i) pairs + random pairs (side-B anchors that occur several times are built once per pair), protein-like density: pairs of
    ~150 ... 350 merged events."""
    rng = np.random.default_rng(100 + n_cat)
    cats = [f"c{i}" for i in range(n_cat)]
    n = 2400
    side = (n / 0.035) ** (1 / 3)
    xa, ca, ta = _cloud(rng, n, side, n_cat)
    xb, cb, tb = _cloud(rng, n - 77, side, n_cat)
    pairs = np.concatenate([np.stack([np.arange(n - 77), np.arange(n - 77)], 1),
                            np.stack([rng.integers(0, n, 3000), rng.integers(0, n - 77, 3000)], 1),
                            np.stack([rng.integers(0, n, 302), np.full(302, 5)], 1)])  # one side-B anchor in 302 pairs; 5625 pairs: not a multiple of 4
    build = lambda mod: mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.1]))
    want = np.asarray(build(oracle).from_arrays(xa, ca, ta, xb, cb, tb, pairs, 9.0))
    regular, rc = _run(lh, monkeypatch, build, xa, ca, ta, xb, cb, tb, pairs, 9.0, repeat=3, per_pair="-1")
    assert rc["per_pair_passes"] == 0 and rc["passes"] == 3
    # without side B's de-duplication: the same environments, the same sweeps -- the same bits
    forced, fc = _run(lh, monkeypatch, build, xa, ca, ta, xb, cb, tb, pairs, 9.0, repeat=3, per_pair="1")
    assert fc["per_pair_passes"] == 3 and fc["passes"] == 3
    for f, r in zip(forced, regular):
        assert np.max(np.abs(f - want)) < TIGHT
        assert np.array_equal(f, r)


@pytest.mark.parametrize("rule", [{"accept_same": False}, {"accept_same": True},
                                  {"tag_pairs": [(0, 1), (2, 2), (5, 3), (7, 7), (1, 6)], "accepted_pairs": True, "ordered": False},
                                  {"tag_pairs": [(0, 1), (2, 2), (5, 3)], "accepted_pairs": False, "ordered": True}])
def test_forced_per_pair_side_b_with_tag_rules_and_weight_functions(lh, oracle, monkeypatch, rule):
    """Both tag-rule instantiations, coarse-grained density (~90 points per environment: four pairs per wavefront), every weight-function
    family as the single function of the configuration."""
    rng = np.random.default_rng(7)
    n_cat = 8
    cats = [f"c{i}" for i in range(n_cat)]
    n = 3000
    side = (n / 0.023) ** (1 / 3)
    listed = "tag_pairs" in rule
    xa, ca, ta = _cloud(rng, n, side, n_cat, 8 if listed else 0)
    xb, cb, tb = _cloud(rng, n, side, n_cat, 8 if listed else 0)
    pairs = np.stack([rng.permutation(n), rng.permutation(n)], 1)
    for wf in (("uniform", [3.0, 10.0]), ("hyper_exp", [1.0, 0.3, 0.1, 0.4]), ("dagum", [2.0, 6.0, 1.5]), ("kumaraswamy", [1.0, 11.0, 1.5, 2.5])):
        def build(mod):
            r = dict(rule)
            if listed:
                r["tag_pairs"] = [(str(x), str(y)) for x, y in rule["tag_pairs"]]
            return mod.LoCoHD(cats, mod.WeightFunction(*wf), mod.TagPairingRule(r))

        if listed:  # interned tags of the arrays are the strings "0" .. "7" of the rule
            import loco_hd_amd.api as api  # noqa: F401  (string tags travel through the list-of-PrimitiveAtom call)
            pa = lambda mod, x, c, t: [mod.PrimitiveAtom(cats[k], str(tt), xyz) for k, tt, xyz in zip(c, t, x)]
            want = np.asarray(build(oracle).from_primitives(pa(oracle, xa, ca, ta), pa(oracle, xb, cb, tb), [tuple(map(int, p)) for p in pairs], 10.0))
            monkeypatch.setenv("LCHD_PER_PAIR", "1")
            lchd = build(lh)
            got = np.asarray(lchd.from_primitives(pa(lh, xa, ca, ta), pa(lh, xb, cb, tb), [tuple(map(int, p)) for p in pairs], 10.0))
            monkeypatch.delenv("LCHD_PER_PAIR")
            assert np.max(np.abs(got - want)) < TIGHT, wf
        else:
            want = np.asarray(build(oracle).from_arrays(xa, ca, ta, xb, cb, tb, pairs, 10.0))
            got, counts = _run(lh, monkeypatch, build, xa, ca, ta, xb, cb, tb, pairs, 10.0, repeat=1, per_pair="1")
            assert counts["per_pair_passes"] == 1
            assert np.max(np.abs(got[0] - want)) < TIGHT, wf


def test_side_b_is_not_deduplicated_when_its_anchors_are_used_once(lh, oracle, monkeypatch):
    """No hook: the first call of a session is a regular pass with de-duplication (it counts the unique anchors); with every side-B
    anchor unique the next calls give every pair its own side-B slot -- the same bits --, a list whose side-B anchors are shared is
    still scored correctly in that mode."""
    rng = np.random.default_rng(11)
    n_cat = 8
    cats = [f"c{i}" for i in range(n_cat)]
    n = 9000
    side = (n / 0.023) ** (1 / 3)
    xa, ca, ta = _cloud(rng, n, side, n_cat)
    xb, cb, tb = _cloud(rng, n, side, n_cat)
    once = np.stack([rng.integers(0, 600, n), rng.permutation(n)], 1)          # side A shared (600 anchors), side B used once
    shared = np.stack([rng.integers(0, n, n), rng.integers(0, 300, n)], 1)     # side B: 300 anchors in 9000 pairs
    build = lambda mod: mod.LoCoHD(cats, mod.WeightFunction("uniform", [3.0, 10.0]), mod.TagPairingRule({"accept_same": False}))
    lo = build(oracle)
    want_once = np.asarray(lo.from_arrays(xa, ca, ta, xb, cb, tb, once, 10.0))
    want_shared = np.asarray(lo.from_arrays(xa, ca, ta, xb, cb, tb, shared, 10.0))
    import torch
    from loco_hd_amd.device import DeviceSession

    monkeypatch.delenv("LCHD_PER_PAIR", raising=False)
    sess = DeviceSession(build(lh))
    a, b = sess.upload(xa, ca, ta), sess.upload(xb, cb, tb)
    d_once, d_shared = torch.from_numpy(once).cuda(), torch.from_numpy(shared).cuda()
    first = sess.from_primitives(a, b, d_once, 10.0).cpu().numpy()
    assert sess.pass_counts()["per_pair_passes"] == 0
    later = [sess.from_primitives(a, b, d_once, 10.0).cpu().numpy() for _ in range(3)]
    assert sess.pass_counts()["per_pair_passes"] == 3
    for g in later:
        assert np.array_equal(g, first)
        assert np.max(np.abs(g - want_once)) < TIGHT
    # shared side-B anchors: the per-pair pass that meets them is still correct; the periodic regular pass would switch it off -- here the
    # caller's next REGULAR pass does (the pass's scoreboard counted the repeats)
    before = sess.pass_counts()["per_pair_passes"]
    assert np.max(np.abs(sess.from_primitives(a, b, d_shared, 10.0).cpu().numpy() - want_shared)) < TIGHT
    assert sess.pass_counts()["per_pair_passes"] == before + 1  # (a list like the hinted one: one pass per pair -- whose scoreboard counts the repeats)
    assert np.max(np.abs(sess.from_primitives(a, b, d_shared, 10.0).cpu().numpy() - want_shared)) < TIGHT
    assert sess.pass_counts()["per_pair_passes"] == before + 1  # ... and the next pass is a regular one again
    # a list with more pairs than side B has atoms repeats anchors by counting: never without de-duplication, whatever came before
    assert np.array_equal(sess.from_primitives(a, b, d_once, 10.0).cpu().numpy(), first)       # (regular: the hint comes back)
    assert np.array_equal(sess.from_primitives(a, b, d_once, 10.0).cpu().numpy(), first)       # (per pair again)
    many = torch.from_numpy(np.concatenate([once, shared, once[::-1]])).cuda()                # 27 000 pairs over 9 000 atoms
    n_pp = sess.pass_counts()["per_pair_passes"]
    got_many = sess.from_primitives(a, b, many, 10.0).cpu().numpy()
    assert sess.pass_counts()["per_pair_passes"] == n_pp
    assert np.max(np.abs(got_many[:n] - want_once)) < TIGHT and np.max(np.abs(got_many[n:2 * n] - want_shared)) < TIGHT
    sess.close()


def test_per_pair_side_b_errors_and_fallbacks(lh, oracle, monkeypatch):
    """An anchor outside its structure raises like the reference's index panic; an environment beyond its slot (a dense cluster) repeats
    the pass with larger slots; categories outside the map raise."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(5)
    n_cat = 6
    cats = [f"c{i}" for i in range(n_cat)]
    n = 5000
    side = (n / 0.02) ** (1 / 3)
    xa, ca, ta = _cloud(rng, n, side, n_cat)
    xb, cb, tb = _cloud(rng, n, side, n_cat)
    v = rng.normal(0, 1, (900, 3))
    xb[:900] = xb[1000] + v / np.linalg.norm(v, axis=1)[:, None] * (4.0 * rng.uniform(0, 1, (900, 1)) ** (1 / 3))  # a 900-point cluster
    pairs = np.stack([rng.permutation(n), rng.permutation(n)], 1)
    build = lambda mod: mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.15]))
    want = np.asarray(build(oracle).from_arrays(xa, ca, np.zeros_like(ta), xb, cb, np.zeros_like(tb), pairs, 8.0))
    got, counts = _run(lh, monkeypatch, build, xa, ca, np.zeros_like(ta), xb, cb, np.zeros_like(tb), pairs, 8.0, per_pair="1")
    assert np.max(np.abs(got[0] - want)) < TIGHT and np.array_equal(got[0], got[1])
    assert counts["passes"] > 2  # the overflow repeated a pass
    monkeypatch.setenv("LCHD_PER_PAIR", "1")
    sess = DeviceSession(build(lh))
    monkeypatch.delenv("LCHD_PER_PAIR")
    a, b = sess.upload(xa[1000:], ca[1000:], np.zeros(n - 1000, np.int32)), sess.upload(xb[1000:], cb[1000:], np.zeros(n - 1000, np.int32))
    bad = np.stack([np.arange(4000), np.arange(4000)], 1)
    bad[3999, 1] = 4000
    with pytest.raises(lh.PanicException):
        sess.from_primitives(a, b, torch.from_numpy(bad).cuda(), 8.0)
    bad[3999, 1] = 17
    ok = sess.from_primitives(a, b, torch.from_numpy(bad).cuda(), 8.0).cpu().numpy()  # the session keeps working
    ref = np.asarray(build(oracle).from_arrays(xa[1000:], ca[1000:], np.zeros(n - 1000, np.int32), xb[1000:], cb[1000:], np.zeros(n - 1000, np.int32), bad, 8.0))
    assert np.max(np.abs(ok - ref)) < TIGHT
    cb_bad = cb[1000:].copy()
    cb_bad[123] = n_cat + 3  # a category outside the map (pmf.rs:38-42: "Category not found!" for every environment that holds the atom)
    b2 = sess.upload(xb[1000:], cb_bad, np.zeros(n - 1000, np.int32))
    with pytest.raises(ValueError):
        sess.from_primitives(a, b2, torch.from_numpy(bad).cuda(), 8.0)
    sess.close()
