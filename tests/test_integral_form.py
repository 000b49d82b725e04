"""The oracle against an independently SHAPED computation of the same score (tests/integral_form.py: the integral
S = sum_j dF_j H_j over the distinct breakpoints with prefix counts, vectorised NumPy in np.longdouble) -- on the
reference's two input collections (every weight function / statistical distance its generator drew) and on random
configurations with category weights, tag rules and from_coords, none of which the reference's own known answers pin.
CPU only; no GPU, no product code."""
import numpy as np
import pytest

import integral_form as iform
from golden_util import load_cases, load_inputs

TOL = 1e-12
CASES = load_cases()


def _oracle_case(oracle, case, meta, seqs, xyz, anchors):
    types = meta["primitive_types"]
    sdp = meta["statistical_distances"][case["sd"]] if case["sd"] is not None else ("Hellinger", [2.0])
    wfp = meta["weight_functions"][case["wf"]]
    lchd = oracle.LoCoHD(types, oracle.WeightFunction(*wfp), statistical_distance=oracle.StatisticalDistance(*sdp))
    i, j = case["i"], case["j"]
    tag = lambda m: np.zeros(m, dtype=np.int32)
    got = np.asarray(lchd.from_arrays(xyz[i], seqs[i], tag(len(seqs[i])), xyz[j], seqs[j], tag(len(seqs[j])), anchors,
                                      meta["threshold_distance"]))
    want = iform.from_primitives(seqs[i], xyz[i], None, seqs[j], xyz[j], None, anchors, meta["threshold_distance"], len(types), wfp, sdp)
    return got, want


@pytest.mark.parametrize("stamp", ["241212_131752", "250605_114749"])
def test_oracle_equals_integral_form_on_reference_inputs(oracle, stamp):
    meta, seqs, xyz = load_inputs(stamp)
    worst, n = 0.0, 0
    for k, case in enumerate(c for c in CASES if c["collection"] == stamp):
        m = min(len(seqs[case["i"]]), len(seqs[case["j"]]))
        anchors = np.asarray([(x, x) for x in range(k % 5, m, 5)], dtype=np.int64)  # a spread of anchors of every case
        got, want = _oracle_case(oracle, case, meta, seqs, xyz, anchors)
        ok = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), ok)
        worst = max(worst, float(np.max(np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok])))))
        n += len(anchors)
    assert n > 1000 and worst < TOL, (n, worst)  # (observed: ~1e-15)


WFS = [("uniform", [3.0, 10.0]), ("hyper_exp", [1.0, 0.1]), ("hyper_exp", [0.49, 0.86, 0.55, 0.13, 0.096, 0.157]),
       ("dagum", [1.7, 2.5, 9.0]), ("kumaraswamy", [2.0, 11.0, 3.3, 4.4])]
SDS = [("Hellinger", [2.0]), ("Hellinger", [3.4277149325231795]), ("Kolmogorov-Smirnov", []), ("Kullback-Leibler", [0.514]),
       ("Renyi", [2.428, 0.73]), ("Renyi", [0.0404, 3.566]), ("Renyi", [1.0, 0.5]), ("Renyi", [float("inf"), 0.25])]


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_with_category_weights_tags_and_dense_rows(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(2, 12))
    cats = [f"c{k}" for k in range(C)]
    wf, sd = WFS[seed % len(WFS)], SDS[(seed * 5 + 1) % len(SDS)]
    weights = None if seed % 3 == 0 else rng.uniform(0.2, 3.0, C).tolist()
    accept_same = bool(seed % 2)
    na, nb = int(rng.integers(40, 160)), int(rng.integers(40, 160))
    lattice = seed % 4 == 1  # integer coordinates: exact distance ties inside and across the two environments
    draw = (lambda n: rng.integers(-4, 5, (n, 3)).astype(float)) if lattice else (lambda n: rng.uniform(-9, 9, (n, 3)))
    xa, xb = draw(na), draw(nb)
    ca, cb = rng.integers(0, C, na).astype(np.int32), rng.integers(0, C, nb).astype(np.int32)
    ta, tb = rng.integers(0, 6, na).astype(np.int32), rng.integers(0, 6, nb).astype(np.int32)
    pairs = np.stack([rng.integers(0, na, 60), rng.integers(0, nb, 60)], 1).astype(np.int64)
    thr = 3.0 if lattice else 7.5
    lchd = oracle.LoCoHD(cats, oracle.WeightFunction(*wf), oracle.TagPairingRule({"accept_same": accept_same}),
                         category_weights=weights, statistical_distance=oracle.StatisticalDistance(*sd))
    got = np.asarray(lchd.from_arrays(xa, ca, ta, xb, cb, tb, pairs, thr))
    want = iform.from_primitives(ca, xa, ta, cb, xb, tb, pairs, thr, C, wf, sd, weights, accept_same)
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert np.max(np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))) < TOL
    # from_coords: every point an anchor, whole structure as environment (nothing in the reference's tests pins it)
    n = min(na, nb, 70)
    lc = oracle.LoCoHD(cats, oracle.WeightFunction(*wf), category_weights=weights, statistical_distance=oracle.StatisticalDistance(*sd))
    got_c = np.asarray(lc.from_coords([cats[k] for k in ca[:n]], [cats[k] for k in cb[:n]], xa[:n], xb[:n]))
    want_c = iform.from_coords(ca[:n], xa[:n], cb[:n], xb[:n], C, wf, sd, weights)
    ok = np.isfinite(want_c)
    assert np.array_equal(np.isfinite(got_c), ok)
    assert np.max(np.abs(got_c[ok] - want_c[ok]) / np.maximum(1.0, np.abs(want_c[ok]))) < TOL


def test_integral_form_reproduces_the_reference_known_answers():
    """The checker itself against the reference's hand-computed values (/root/reference/tests/test_locohd.py:27-52)."""
    seq = [0, 1, 2, 3]
    v = iform.score(seq, [0.0, 1.0, 2.0, 3.0], seq, [0.0, 1.0, 1.0, 1.0], 4, ("uniform", [0.0, 4.0]))
    assert v == pytest.approx(0.2268, abs=5e-5)
    v = iform.score([0, 1, 0, 2], [0.0, 1.0, 5.0, 9.0], [0, 2], [0.0, 7.0], 3, ("kumaraswamy", [3.0, 10.0, 2.0, 5.0]))
    assert v == pytest.approx(0.4979, abs=5e-5)
