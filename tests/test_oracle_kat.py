"""Pins the CPU oracle against every known-answer the reference's own tests hold for the path.

Values and inputs are those of /root/reference/tests/test_locohd.py:27-73,
tests/test_tag_pairing_rule.py:8-157 and tests/test_wfs.py:8-156 (hand-calculated by the reference
authors); nothing here needs a GPU.
"""
import itertools

import pytest


def test_small_locohd(oracle):  # tests/test_locohd.py:27-52
    lchd = oracle.LoCoHD(["O", "A", "B", "C"], oracle.WeightFunction("uniform", [0.0, 4.0]))
    seq = ["O", "A", "B", "C"]
    assert lchd.from_anchors(seq, seq, [0.0, 1.0, 2.0, 3.0], [0.0, 1.0, 1.0, 1.0]) == pytest.approx(0.2268, abs=5e-5)
    assert lchd.from_anchors(seq, seq, [0.0, 1.0, 1.0, 1.0], [0.0, 1.0, 2.0, 3.0]) == pytest.approx(0.2268, abs=5e-5)
    lchd = oracle.LoCoHD(["A", "B", "C"], oracle.WeightFunction("kumaraswamy", [3.0, 10.0, 2.0, 5.0]))
    v = lchd.from_anchors(["A", "B", "A", "C"], ["A", "C"], [0.0, 1.0, 5.0, 9.0], [0.0, 7.0])
    assert v == pytest.approx(0.4979, abs=5e-5)


def test_locohd_ctor_errors(oracle):  # tests/test_locohd.py:54-73
    wf = oracle.WeightFunction("uniform", [0.0, 4.0])
    types = ["O", "A", "B", "C"]
    with pytest.raises(ValueError):
        oracle.LoCoHD([], wf)
    for bad in ([1.0, 1.0, 1.0], [1.0] * 5, [1.0, -1.0, 1.0, 1.0], [1.0, 0.0, 1.0, 1.0]):
        with pytest.raises(ValueError):
            oracle.LoCoHD(types, wf, category_weights=bad)


WF_KATS = [  # tests/test_wfs.py:8-138
    ("hyper_exp", [1.0, 1.0], [(0, 1, 0.6321), (1, 3, 0.3181), (5, 10, 0.0067)]),
    ("hyper_exp", [0.5, 0.5, 1 / 2, 1 / 3], [(0, 1, 0.3385), (1, 3, 0.3660), (5, 10, 0.1143)]),
    ("hyper_exp", [3.0, 5.0, 2.0, 1 / 3, 1 / 5, 1 / 10], [(0, 1, 0.1947), (1, 3, 0.2724), (5, 10, 0.2100)]),
    ("dagum", [1.0, 1.0, 1.0], [(0, 1, 0.5), (1, 3, 0.25), (5, 10, 0.0758)]),
    ("dagum", [2.0, 5.0, 1.0], [(0, 1, 0.0385), (1, 3, 0.2262), (5, 10, 0.3)]),
    ("dagum", [10.0, 5.0, 2.0], [(0, 1, 0.0), (1, 3, 0.0), (5, 10, 0.7480)]),
    ("uniform", [0.0, 1.0], [(0, 1, 1.0), (1, 3, 0.0), (5, 10, 0.0)]),
    ("uniform", [3.0, 10.0], [(0, 1, 0.0), (1, 3, 0.0), (5, 10, 0.7143)]),
    ("uniform", [2.0, 16.0], [(0, 1, 0.0), (1, 3, 0.0714), (5, 10, 0.3571)]),
    ("kumaraswamy", [1.0, 2.0, 2.0, 2.0], [(1.0, 2.0, 1.0), (1.25, 1.75, 0.6875), (1.4, 10.0, 0.7056)]),
    ("kumaraswamy", [5.0, 10.0, 2.0, 3.0], [(5, 7, 0.4073), (1, 17, 1.0), (6.4, 6.7, 0.0910)]),
    ("kumaraswamy", [5.0, 9.0, 7.0, 7.0], [(5, 7, 0.0534), (1, 17, 1.0), (6.4, 6.7, 0.0129)]),
]
WF_ERRS = [  # tests/test_wfs.py:29-156
    ("hyper_exp", [1.0]), ("hyper_exp", [1.0, 2.0, 3.0]), ("hyper_exp", [-1.0, 1.0]), ("hyper_exp", [1.0, -1.0]),
    ("hyper_exp", [1.0, -1.0, 2.0]), ("dagum", [1.0]), ("dagum", [1.0, 2.0]), ("dagum", [-1.0, 2.0, 3.0]),
    ("dagum", [1.0, -2.0, 3.0]), ("dagum", [1.0, 2.0, -3.0]), ("uniform", [1.0]), ("uniform", [1.0, 0.0]),
    ("uniform", [-1.0, 0.0]), ("kumaraswamy", [1.0]), ("kumaraswamy", [1.0, 2.0]), ("kumaraswamy", [3.0, 1.0, 2.0, 2.0]),
    ("kumaraswamy", [1.0, 3.0, -2.0, 2.0]), ("kumaraswamy", [0.0, 3.0, 2.0, -2.0]),
]


@pytest.mark.parametrize("name,params,cases", WF_KATS)
def test_weight_function_kats(oracle, name, params, cases):
    wf = oracle.WeightFunction(name, params)
    for a, b, want in cases:
        assert wf.integral_range(float(a), float(b)) == pytest.approx(want, abs=5e-5)


@pytest.mark.parametrize("name,params", WF_ERRS)
def test_weight_function_errors(oracle, name, params):
    with pytest.raises(ValueError):
        oracle.WeightFunction(name, params)


def test_tag_pairing_truth_table(oracle):  # tests/test_tag_pairing_rule.py:8-98
    tpr = oracle.TagPairingRule({"accept_same": True})
    assert tpr.pair_accepted(("A", "A")) and not tpr.pair_accepted(("A", "B"))
    tpr = oracle.TagPairingRule({"accept_same": False})
    assert not tpr.pair_accepted(("A", "A")) and tpr.pair_accepted(("A", "B"))
    listed = {("A", "B"), ("A", "C"), ("B", "C")}
    for accepted_pairs, ordered in itertools.product([True, False], [True, False]):
        tpr = oracle.TagPairingRule({"tag_pairs": listed, "accepted_pairs": accepted_pairs, "ordered": ordered})
        for pair in itertools.product("ABC", repeat=2):
            hit = pair in listed or (not ordered and pair[::-1] in listed)
            assert tpr.pair_accepted(pair) == (hit if accepted_pairs else not hit), (accepted_pairs, ordered, pair)


def planar_structure(mod):  # tests/test_tag_pairing_rule.py:104-118
    pts = [("A", [0, 0, 0]), ("A", [0, 1, 0]), ("A", [2, 0, 0]), ("A", [2, 2, 0]), ("B", [1, 2, 0]), ("B", [1, 3, 0]),
           ("B", [3, 2, 0]), ("B", [3, 3, 0]), ("C", [2, 1, 0])]
    return [mod.PrimitiveAtom(t, t, [float(x) for x in c]) for t, c in pts]


PLANAR_ANCHORS = [(0, 3), (4, 5), (0, 4), (0, 8), (4, 8)]


def test_tag_rule_in_locohd(oracle):  # tests/test_tag_pairing_rule.py:100-157
    s = planar_structure(oracle)
    wf = oracle.WeightFunction("uniform", [1.0, 1.001])
    lchd = oracle.LoCoHD(["A", "B", "C"], wf, oracle.TagPairingRule({"accept_same": True}))
    got = lchd.from_primitives(s, s, PLANAR_ANCHORS, 1.002)
    for g, w in zip(got, [0.0, 0.0, 1.0, 1.0, 1.0]):
        assert abs(g - w) < 5e-16
    lchd = oracle.LoCoHD(["A", "B", "C"], wf, oracle.TagPairingRule({"accept_same": False}))
    got = lchd.from_primitives(s, s, PLANAR_ANCHORS, 1.002)
    for g, w in zip(got, [0.7071, 0.5412, 0.5412, 0.4284, 0.6501]):
        assert g == pytest.approx(w, abs=5e-5)
