"""Kullback-Leibler and Renyi divergences with a tiny smoothing constant take an O(1)-per-event sweep (k_sweep_inc,
loco_hd_amd/csrc/lchd_sweep_inc.hip) instead of the generic per-category evaluation; reference:
/root/reference/src/locohd/pmf/statistical_distances.rs:23-78.  Both paths against the CPU oracle and against each other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT = 1e-11
CATS = [f"c{i}" for i in range(10)]


@pytest.fixture(scope="module")
def lh():
    import loco_hd_amd

    return loco_hd_amd


def prims(mod, seq, xyz, tags=None):
    tags = [""] * len(seq) if tags is None else tags
    return [mod.PrimitiveAtom(s, t, c) for s, t, c in zip(seq, tags, xyz)]


CASES = [("Kullback-Leibler", [1e-10]), ("Kullback-Leibler", [1e-9]), ("Kullback-Leibler", [1e-14]), ("Renyi", [1.0, 1e-10]),
         ("Renyi", [2.4, 1e-10]), ("Renyi", [0.5, 1e-10]), ("Renyi", [0.05, 1e-12]), ("Renyi", [1.3, 1e-9]), ("Renyi", [6.0, 1e-11]),
         # outside the fast path's conditions (generic sweep): a larger eps, the special orders, an order beyond the tables
         ("Kullback-Leibler", [1e-6]), ("Renyi", [0.0, 1e-10]), ("Renyi", [float("inf"), 1e-10]), ("Renyi", [25.0, 1e-10])]


@pytest.mark.parametrize("sd", CASES)
def test_fast_divergences_match_the_oracle(lh, oracle, sd, monkeypatch):
    rng = np.random.default_rng(81)
    # protein-like density (environments of ~170 points), a sparse structure (most categories missing on one side for most of
    # the sweep: the b = 0 class), few categories, and a structure compared with a copy of itself
    setups = [(900, 26.0, 10.0, CATS), (300, 30.0, 9.0, CATS), (500, 20.0, 8.0, CATS[:3])]
    for n, side, thr, cats in setups:
        sa, xa = rng.choice(cats, n).tolist(), rng.uniform(0, side, (n, 3))
        sb, xb = rng.choice(cats, n).tolist(), rng.uniform(0, side, (n, 3))
        anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, n, 500), rng.integers(0, n, 500))]
        for (qa, ya, qb, yb) in ((sa, xa, sb, xb), (sa, xa, sa, xa)):

            def run(mod):
                lchd = mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.12]), statistical_distance=mod.StatisticalDistance(*sd))
                return [np.asarray(lchd.from_primitives(prims(mod, qa, ya), prims(mod, qb, yb), anchors, thr)) for _ in range(2)]

            got, want = run(lh), run(oracle)[0]
            scale = max(1.0, float(np.max(np.abs(want[np.isfinite(want)]))) if np.isfinite(want).any() else 1.0)
            for g in got:
                assert np.array_equal(np.isfinite(g), np.isfinite(want))
                ok = np.isfinite(want)
                assert np.max(np.abs(g[ok] - want[ok])) < TIGHT * scale, (sd, n)
            monkeypatch.setenv("LCHD_NO_SD_INC", "1")
            slow = run(lh)[0]
            monkeypatch.delenv("LCHD_NO_SD_INC")
            ok = np.isfinite(want)
            assert np.max(np.abs(slow[ok] - got[0][ok])) < TIGHT * scale, (sd, n)


def test_fast_divergences_fall_back_where_they_must(lh, oracle):
    """Category weights, a weight-function dictionary, environments beyond 512 points and a tag rule: the first three leave the
    fast path's conditions (generic sweep), the tag rule stays on it."""
    rng = np.random.default_rng(82)
    n = 700
    sa, xa = rng.choice(CATS, n).tolist(), rng.uniform(0, 24.0, (n, 3))
    sb, xb = rng.choice(CATS, n).tolist(), rng.uniform(0, 24.0, (n, 3))
    tags = [f"r{i // 4}" for i in range(n)]
    anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, n, 300), rng.integers(0, n, 300))]
    sd = ("Renyi", [2.0, 1e-10])
    variants = [dict(category_weights=list(np.linspace(0.5, 2.0, 10))), dict(thr=40.0), dict(rule={"accept_same": False}),
                dict(w_func="dict")]
    for v in variants:

        def run(mod):
            kw = dict(statistical_distance=mod.StatisticalDistance(*sd))
            if "category_weights" in v:
                kw["category_weights"] = v["category_weights"]
            wf = mod.WeightFunction("uniform", [1.0, 9.0])
            keyed = anchors
            if v.get("w_func") == "dict":
                wf = {"u": mod.WeightFunction("uniform", [1.0, 9.0]), "h": mod.WeightFunction("hyper_exp", [1.0, 0.2])}
                keyed = [(i, j, "u" if i % 2 else "h") for i, j in anchors]
            rule = mod.TagPairingRule(v["rule"]) if "rule" in v else None
            lchd = mod.LoCoHD(CATS, wf, rule, **kw)
            return np.asarray(lchd.from_primitives(prims(mod, sa, xa, tags), prims(mod, sb, xb, tags), keyed, v.get("thr", 9.0)))

        got, want = run(lh), run(oracle)
        scale = max(1.0, float(np.max(np.abs(want))))
        assert np.max(np.abs(got - want)) < TIGHT * scale, v
