"""The reference's import name: `from loco_hd import ...` (/root/reference/loco_hd/__init__.py:1-2) resolves to this build."""
import numpy as np
import pytest


def test_import_names_are_the_hip_build():
    import loco_hd
    import loco_hd_amd
    from loco_hd import (LoCoHD, PrimitiveAssigner, PrimitiveAtom, PrimitiveAtomSource, PrimitiveAtomTemplate, StatisticalDistance,
                         TagPairingRule, TypingSchemeElement, WeightFunction)
    from loco_hd.loco_hd import LoCoHD as core_locohd  # the extension module's name (src/lib.rs:9-17)

    for name, obj in (("LoCoHD", LoCoHD), ("PrimitiveAtom", PrimitiveAtom), ("WeightFunction", WeightFunction),
                      ("TagPairingRule", TagPairingRule), ("StatisticalDistance", StatisticalDistance),
                      ("PrimitiveAssigner", PrimitiveAssigner), ("PrimitiveAtomTemplate", PrimitiveAtomTemplate),
                      ("PrimitiveAtomSource", PrimitiveAtomSource), ("TypingSchemeElement", TypingSchemeElement)):
        assert obj is getattr(loco_hd_amd, name)
    assert core_locohd is loco_hd_amd.LoCoHD
    assert "reference" not in (loco_hd.__file__ or "")
    # host-side leaves work without a GPU (tests/test_wfs.py of the reference: uniform[3,10] at 6.5)
    assert WeightFunction("uniform", [3.0, 10.0]).integral_point(6.5) == 0.5
    assert TagPairingRule({"accept_same": False}).pair_accepted(("a", "b"))


@pytest.mark.gpu
def test_simple_test_call_pattern(oracle):
    """python_codes/simple_test.py:9-27 verbatim in shape: NumPy arrays of str labels and of points, positional arguments."""
    from loco_hd import LoCoHD, WeightFunction

    rng = np.random.default_rng(5)
    categories = list(map(str, range(6)))
    n_of_points = 100
    weight_function = WeightFunction("hyper_exp", [1., 1.])
    lchd = LoCoHD(categories, weight_function)
    ref = oracle.LoCoHD(categories, oracle.WeightFunction("hyper_exp", [1., 1.]))
    points = rng.uniform(0., 1., size=(n_of_points, 3))
    sequence1 = rng.choice(categories, size=n_of_points)
    for delta in [0., 1., 2., 4., 8., 16., 32., 64.]:
        directions = rng.normal(0, 1, size=(n_of_points, 3))
        normals = np.sqrt(np.sum(directions ** 2, axis=1, keepdims=True))
        directions = points + delta * directions / normals
        sequence2 = rng.choice(categories, size=n_of_points)
        lchd_score = lchd.from_coords(sequence1, sequence2, points, directions)
        assert isinstance(lchd_score, list) and len(lchd_score) == n_of_points
        want = ref.from_coords(list(sequence1), list(sequence2), points, directions)
        assert np.max(np.abs(np.asarray(lchd_score) - np.asarray(want))) < 1e-11
