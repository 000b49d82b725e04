"""Golden fixtures: the reference's own random test inputs (tests/test_data input pickles, repacked) with
expected (mean, median, std, min, max) per case -- the shape of the reference's consistency test
(/root/reference/tests/test_locohd.py:75-133).  Expected values are oracle-generated (the reference's output
pickles are absent, see tests/golden/make_golden.py).  CPU: the oracle must keep reproducing them bit-for-bit
(places=15 like the reference).  GPU: the HIP path must match them within 1e-6 (observed ~1e-15)."""
import numpy as np
import pytest

from golden_util import load_cases, run_case, stats

CASES = load_cases()


def test_fixture_shape():
    assert len(CASES) == 296
    assert sum(1 for c in CASES if c["sd"] is None) == 40 and sum(1 for c in CASES if c["sd"] is not None) == 256


@pytest.mark.parametrize("k", range(0, len(CASES), 7))
def test_oracle_reproduces_golden(oracle, k):
    case = CASES[k]
    scores = run_case(oracle, case)
    for got, want in zip(stats(scores), case["stats"]):
        assert abs(got - want) < 5e-16 * max(1.0, abs(want))
    if "scores" in case:
        assert np.array_equal(scores, np.asarray(case["scores"]))


@pytest.mark.gpu
def test_hip_matches_golden():
    import loco_hd_amd as lh

    worst = 0.0
    for case in CASES:
        scores = run_case(lh, case)
        assert np.all(np.isfinite(scores))
        worst = max(worst, max(abs(g - w) for g, w in zip(stats(scores), case["stats"])))
        if "scores" in case:
            worst = max(worst, float(np.max(np.abs(scores - np.asarray(case["scores"])))))
    assert worst < 1e-10, worst
