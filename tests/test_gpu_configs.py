"""BASELINE.json configs 3-5 (and the dense variant of config 2) as parity cases: the synthetic inputs of
SURVEY.md section 8(d) at sizes the CPU oracle finishes in seconds, plus size-independent properties at the full
sizes (self-comparison = 0, symmetry, range, anchor order).  The full-size C2a / C5 runs with an oracle parity
gate over ~10^5-10^6 pairs live in bench.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT = 1e-11
CG_TYPES = ["Cent", "AmideC", "OH", "Pos", "Neg", "Aro", "Ali", "Sulf"]  # primitive_typings/coarse_grained_with_centroid.config.json


@pytest.fixture(scope="module")
def lh():
    import loco_hd_amd

    return loco_hd_amd


def cg_structure(rng, n_res, side):
    """3 points per residue, the first one a "Cent" centroid; tag = per-residue id."""
    n = 3 * n_res
    xyz = rng.uniform(0.0, side, (n, 3))
    seq = [CG_TYPES[0] if i % 3 == 0 else CG_TYPES[1 + int(k)] for i, k in enumerate(rng.integers(0, 7, n))]
    tags = [f"A/{i // 3}-RES" for i in range(n)]
    return seq, xyz, tags


def prims(mod, seq, xyz, tags):
    return [mod.PrimitiveAtom(s, t, c) for s, t, c in zip(seq, tags, xyz)]


def test_c3_casp_style_all_vs_all(lh, oracle):
    """Decoy-vs-decoy with centroid anchors, accept_same=False, uniform[3,10], threshold 10
    (python_codes/casp14/casp14_extend_with_locohd.py:35-37,72-79)."""
    rng = np.random.default_rng(3)
    n_res = 200
    side = (3 * n_res / 0.023) ** (1 / 3)
    base = cg_structure(rng, n_res, side)
    decoys = [(base[0], base[1] + rng.normal(0.0, 1.5, base[1].shape), base[2]) for _ in range(5)]
    anchors = [(i, i) for i in range(0, 3 * n_res, 3)]

    def run(mod):
        lchd = mod.LoCoHD(CG_TYPES, mod.WeightFunction("uniform", [3.0, 10.0]), mod.TagPairingRule({"accept_same": False}))
        ps = [prims(mod, *d) for d in decoys]
        return {(a, b): np.asarray(lchd.from_primitives(ps[a], ps[b], anchors, 10.0)) for a in range(5) for b in range(a + 1, 5)}

    got, want = run(lh), run(oracle)
    for k in want:
        assert np.max(np.abs(got[k] - want[k])) < TIGHT, k
    # symmetry of the Hellinger-based score under swapping the two structures
    lchd = lh.LoCoHD(CG_TYPES, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule({"accept_same": False}))
    ab = np.asarray(lchd.from_primitives(prims(lh, *decoys[0]), prims(lh, *decoys[1]), anchors, 10.0))
    ba = np.asarray(lchd.from_primitives(prims(lh, *decoys[1]), prims(lh, *decoys[0]), anchors, 10.0))
    assert np.max(np.abs(ab - ba)) < 1e-13


def test_c4_md_trajectory_frames(lh, oracle):
    """One reference structure vs jittered frames through the device-resident session
    (python_codes/trajectory_analyzer.py:97-119): frames only replace coordinates."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(4)
    n_res = 220
    side = (3 * n_res / 0.023) ** (1 / 3)
    seq, xyz, tags = cg_structure(rng, n_res, side)
    frames = [xyz + rng.normal(0.0, 0.5, xyz.shape) for _ in range(6)]
    anchors = [(i, i) for i in range(0, 3 * n_res, 3)]
    rule = {"accept_same": False}

    lchd = lh.LoCoHD(CG_TYPES, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule(rule))
    interner = {}
    packed = lchd.pack(prims(lh, seq, xyz, tags), interner)
    sess = DeviceSession(lchd, interner=interner)
    ref = sess.upload(packed.xyz, packed.cat, packed.tag)
    cur = sess.upload(packed.xyz, packed.cat, packed.tag)
    d_anchors = torch.tensor(anchors, dtype=torch.int64, device="cuda")
    got = []
    for f in frames:
        sess.set_coords(cur, f)
        got.append(sess.from_primitives(ref, cur, d_anchors, 10.0).cpu().numpy())
    sess.close()

    lo = oracle.LoCoHD(CG_TYPES, oracle.WeightFunction("uniform", [3.0, 10.0]), oracle.TagPairingRule(rule))
    ref_p = prims(oracle, seq, xyz, tags)
    for f, g in zip(frames, got):
        want = np.asarray(lo.from_primitives(ref_p, prims(oracle, seq, f, tags), anchors, 10.0))
        assert np.max(np.abs(g - want)) < TIGHT


def test_c5_stress_reduced(lh, oracle):
    """Two large labelled clouds, 25 categories, random anchor pairs (reduced: 30k points, 40k pairs)."""
    rng = np.random.default_rng(5)
    n, c = 30_000, 25
    side = (n / 0.05) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, c, n).astype(np.int32), rng.integers(0, c, n).astype(np.int32)
    pairs = np.stack([rng.integers(0, n, 40_000), rng.integers(0, n, 40_000)], 1).astype(np.int64)
    cats = [f"c{i}" for i in range(c)]
    tag = np.zeros(n, dtype=np.int32)
    lchd = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1]))
    from loco_hd_amd.api import _Packed

    got = lchd.from_packed(_Packed(xa, ca, tag), _Packed(xb, cb, tag), pairs, 10.0)
    lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.1]))
    want = np.asarray(lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, 10.0))
    assert np.max(np.abs(got - want)) < TIGHT
    assert got.min() >= 0.0 and got.max() <= 1.0 + 1e-12


def test_c2b_dense_from_coords_properties(lh):
    """from_coords on 6000-atom clouds (whole-cloud environments): properties that need no oracle."""
    rng = np.random.default_rng(2)
    n = 6000
    side = (n / 0.05) ** (1 / 3)
    cats = [f"c{i}" for i in range(10)]
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    sa, sb = rng.choice(cats, n).tolist(), rng.choice(cats, n).tolist()
    lchd = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1]))
    same = np.asarray(lchd.from_coords(sa, sa, xa, xa))
    assert np.max(np.abs(same)) == 0.0
    ab, ba = np.asarray(lchd.from_coords(sa, sb, xa, xb)), np.asarray(lchd.from_coords(sb, sa, xb, xa))
    assert np.max(np.abs(ab - ba)) < 1e-13 and ab.min() >= 0.0 and ab.max() <= 1.0
    # a permutation of the atoms permutes the scores (every atom is its own anchor; environments are sets)
    perm = rng.permutation(n)
    pa = np.asarray(lchd.from_coords([sa[i] for i in perm], [sb[i] for i in perm], xa[perm], xb[perm]))
    assert np.max(np.abs(pa - ab[perm])) < 1e-13
    # and the thresholded path with a threshold beyond the cloud diameter sees the same environments
    prim = lambda s, x: [lh.PrimitiveAtom(t, "", c) for t, c in zip(s[:1500], x[:1500])]
    dense = np.asarray(lchd.from_coords(sa[:1500], sb[:1500], xa[:1500], xb[:1500]))
    thr = np.asarray(lchd.from_primitives(prim(sa, xa), prim(sb, xb), [(i, i) for i in range(1500)], 1e4))
    assert np.max(np.abs(dense - thr)) < 1e-13


def test_full_size_c2a_properties(lh):
    """10k-atom clouds, 10^5 anchor pairs: order preservation and repeat-consistency on the device-resident path."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(2)
    n, c = 10_000, 10
    side = (n / 0.05) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, c, n).astype(np.int32), rng.integers(0, c, n).astype(np.int32)
    lchd = lh.LoCoHD([f"c{i}" for i in range(c)], lh.WeightFunction("hyper_exp", [1.0, 0.1]))
    sess = DeviceSession(lchd)
    a, b = sess.upload(xa, ca), sess.upload(xb, cb)
    pairs = torch.from_numpy(np.stack([rng.integers(0, n, 100_000), rng.integers(0, n, 100_000)], 1)).cuda()
    first = sess.from_primitives(a, b, pairs, 10.0).clone()  # (no pair statistics yet: the device picks the sweep kernel)
    s1 = sess.from_primitives(a, b, pairs, 10.0).clone()
    assert float((first - s1).abs().max()) < 1e-14  # another kernel variant = another summation order of the same terms
    perm = torch.randperm(pairs.shape[0], device="cuda")
    s2 = sess.from_primitives(a, b, pairs[perm].contiguous(), 10.0)
    assert torch.equal(s1[perm], s2)  # output i <-> anchor pair i, bit for bit (same launch set: picked from the previous pass)
    s3 = sess.from_primitives(b, a, pairs.flip(1).contiguous(), 10.0)
    assert float((s1 - s3).abs().max()) < 1e-13
    assert float(s1.min()) >= 0.0 and float(s1.max()) <= 1.0
    sess.close()


def test_batched_structures_match_single_calls(lh):
    """C3 / C4 as ONE device call: many structures in one batch object, anchor pairs of many structure pairs in one
    anchor tensor.  Must equal the per-structure-pair calls bit for bit (environments never mix structures)."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(6)
    n_res = 150
    side = (3 * n_res / 0.023) ** (1 / 3)
    base = cg_structure(rng, n_res, side)
    decoys = [(base[0], base[1] + rng.normal(0.0, 1.0 + 0.3 * k, base[1].shape), base[2]) for k in range(6)]
    lchd = lh.LoCoHD(CG_TYPES, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule({"accept_same": False}))
    interner = {}
    packed = [lchd.pack(prims(lh, *d), interner) for d in decoys]
    local_anchors = np.arange(0, 3 * n_res, 3)

    # reference: one from_primitives call per structure pair
    want = {}
    for a in range(6):
        for b in range(a + 1, 6):
            want[(a, b)] = lchd.from_packed(packed[a], packed[b], np.stack([local_anchors, local_anchors], 1), 10.0, interner=interner)

    sess = DeviceSession(lchd, interner=interner)
    batch, offs = sess.upload_batch([(p.xyz, p.cat, p.tag) for p in packed])
    pair_list = sorted(want)
    anchors = np.concatenate([np.stack([offs[a] + local_anchors, offs[b] + local_anchors], 1) for a, b in pair_list])
    scores = sess.from_primitives(batch, batch, torch.from_numpy(anchors).cuda(), 10.0).cpu().numpy()
    for k, (a, b) in enumerate(pair_list):
        got = scores[k * len(local_anchors):(k + 1) * len(local_anchors)]
        assert np.array_equal(got, want[(a, b)]), (a, b)

    # C4 shape: one reference structure against a batch of frames
    ref = sess.upload(packed[0].xyz, packed[0].cat, packed[0].tag)
    frames, foffs = sess.upload_batch([(p.xyz, p.cat, p.tag) for p in packed[1:]])
    anchors = np.concatenate([np.stack([local_anchors, foffs[f] + local_anchors], 1) for f in range(5)])
    scores = sess.from_primitives(ref, frames, torch.from_numpy(anchors).cuda(), 10.0).cpu().numpy()
    for f in range(5):
        assert np.array_equal(scores[f * len(local_anchors):(f + 1) * len(local_anchors)], want[(0, f + 1)])
    sess.close()


def test_trajectory_streaming_matches_single_calls(lh, oracle):
    """BASELINE config 4: frames streamed in chunks through two device buffers on a copy stream."""
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(8)
    n_res = 120
    side = (3 * n_res / 0.023) ** (1 / 3)
    seq, xyz, tags = cg_structure(rng, n_res, side)
    n_frames = 23  # not a multiple of the chunk size
    frames = xyz[None, :, :] + rng.normal(0.0, 0.5, (n_frames,) + xyz.shape)
    la = np.arange(0, 3 * n_res, 3)
    local_pairs = np.stack([la, la], 1)
    rule = {"accept_same": False}
    lchd = lh.LoCoHD(CG_TYPES, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule(rule))
    interner = {}
    packed = lchd.pack(prims(lh, seq, xyz, tags), interner)
    sess = DeviceSession(lchd, interner=interner)
    ref = sess.upload(packed.xyz, packed.cat, packed.tag)
    got = sess.score_trajectory(ref, frames, local_pairs, 10.0, chunk=5)
    assert got.shape == (n_frames, len(la))
    sess.close()
    lo = oracle.LoCoHD(CG_TYPES, oracle.WeightFunction("uniform", [3.0, 10.0]), oracle.TagPairingRule(rule))
    ref_p = prims(oracle, seq, xyz, tags)
    for f in (0, 4, 5, 11, 22):
        want = np.asarray(lo.from_primitives(ref_p, prims(oracle, seq, frames[f], tags), [(int(i), int(i)) for i in la], 10.0))
        assert np.max(np.abs(got[f] - want)) < TIGHT, f
    # non-finite coordinates in a frame are refused
    sess = DeviceSession(lchd, interner=interner)
    ref = sess.upload(packed.xyz, packed.cat, packed.tag)
    bad = frames.copy()
    bad[3, 7, 1] = np.nan
    with pytest.raises(ValueError):
        sess.score_trajectory(ref, bad, local_pairs, 10.0, chunk=8)
    sess.close()


def test_dense_rows_beyond_lds_capacity(lh, oracle):
    """from_coords on 17 000-atom clouds: rows no longer fit in LDS, the keys are bucket-sorted in global memory.
    Checked against the oracle's sweep on NumPy-sorted environments for a sample of rows, plus self-comparison."""
    rng = np.random.default_rng(9)
    n = 17_000
    side = (n / 0.05) ** (1 / 3)
    cats = [f"c{i}" for i in range(6)]
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    sa, sb = rng.choice(cats, n).tolist(), rng.choice(cats, n).tolist()
    lchd = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1]))
    got = np.asarray(lchd.from_coords(sa, sb, xa, xb))
    assert got.shape == (n,) and np.all(np.isfinite(got)) and got.min() >= 0.0 and got.max() <= 1.0
    lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.1]))
    sa_arr, sb_arr = np.asarray(sa), np.asarray(sb)
    for i in (0, 1, 4242, 9999, n - 1):
        da = np.sqrt((((xa[i] - xa) ** 2)[:, 0] + ((xa[i] - xa) ** 2)[:, 1]) + ((xa[i] - xa) ** 2)[:, 2])
        db = np.sqrt((((xb[i] - xb) ** 2)[:, 0] + ((xb[i] - xb) ** 2)[:, 1]) + ((xb[i] - xb) ** 2)[:, 2])
        oa, ob = np.argsort(da, kind="stable"), np.argsort(db, kind="stable")
        want = lo.from_anchors(sa_arr[oa].tolist(), sb_arr[ob].tolist(), da[oa].tolist(), db[ob].tolist())
        assert abs(got[i] - want) < TIGHT, i
    same = np.asarray(lchd.from_coords(sa, sa, xa, xa))
    assert np.max(np.abs(same)) == 0.0


def test_dense_row_kernels_agree(lh, oracle, monkeypatch):
    """The dense-row paths against the oracle and each other: rows in one distance segment (k_env_rows2), rows of 16385 ..
    20480 points in two segments, a long row that defeats the segmented sort (all distances equal: the call is repeated with
    the global-memory row sort), rows beyond 20480 points, ragged / infinite entries, and the older kernel for every length."""
    rng = np.random.default_rng(91)
    cats = [f"c{i}" for i in range(7)]
    wf = ("hyper_exp", [0.7, 0.3, 0.2, 0.05])

    def rows(n_cols, n_rows, kind):
        m = rng.uniform(0.0, 60.0, (n_rows, n_cols)) if kind != "equal" else np.full((n_rows, n_cols), 5.0)
        if kind == "inf":
            m[:, rng.integers(1, n_cols, n_cols // 50)] = np.inf
        m[:, 0] = 0.0
        return m

    for n_cols, n_rows, kind in ((900, 6, "random"), (5000, 4, "inf"), (12_000, 3, "random"), (17_000, 3, "random"), (20_480, 2, "inf"),
                                 (17_000, 2, "equal"), (21_000, 2, "random")):
        sa, sb = rng.choice(cats, n_cols).tolist(), rng.choice(cats, n_cols).tolist()
        ma, mb = rows(n_cols, n_rows, kind), rows(n_cols, n_rows, kind)
        want = np.asarray(oracle.LoCoHD(cats, oracle.WeightFunction(*wf)).from_dmxs(sa, sb, ma.tolist(), mb.tolist()))
        got = np.asarray(lh.LoCoHD(cats, lh.WeightFunction(*wf)).from_dmxs(sa, sb, ma, mb))
        assert np.max(np.abs(got - want)) < TIGHT, (n_cols, kind)
        if n_cols <= 17_000 and kind != "equal":
            monkeypatch.setenv("LCHD_OLD_ROWS", "1")
            old = np.asarray(lh.LoCoHD(cats, lh.WeightFunction(*wf)).from_dmxs(sa, sb, ma, mb))
            monkeypatch.delenv("LCHD_OLD_ROWS")
            assert np.max(np.abs(got - old)) < 1e-13, (n_cols, kind)
    # from_coords, both kernels, a length with a partly filled last wavefront; two weight functions (distance keys, no CDF keys)
    n = 3001
    xa, xb = rng.uniform(0, 40, (n, 3)), rng.uniform(0, 40, (n, 3))
    sa, sb = rng.choice(cats, n).tolist(), rng.choice(cats, n).tolist()
    multi = lambda mod: {"a": mod.WeightFunction(*wf), "b": mod.WeightFunction("uniform", [2.0, 30.0])}
    keys = ["a" if k % 3 else "b" for k in range(n)]
    got = np.asarray(lh.LoCoHD(cats, multi(lh)).from_coords(sa, sb, xa, xb, keys))
    monkeypatch.setenv("LCHD_OLD_ROWS", "1")
    old = np.asarray(lh.LoCoHD(cats, multi(lh)).from_coords(sa, sb, xa, xb, keys))
    monkeypatch.delenv("LCHD_OLD_ROWS")
    assert np.max(np.abs(got - old)) < 1e-13
    sample = list(range(0, n, 211))
    want = np.asarray(oracle.LoCoHD(cats, multi(oracle)).from_coords(sa, sb, xa, xb, keys))[sample]
    assert np.max(np.abs(got[sample] - want)) < TIGHT


def test_from_primitives_batch_matches_single_calls(lh, oracle):
    """LoCoHD.from_primitives_batch (additive): several structure pairs in one device pass, bit-identical to
    the per-pair calls and equal to the oracle."""
    rng = np.random.default_rng(12)
    n_res = 60
    side = (3 * n_res / 0.023) ** (1 / 3)
    base = cg_structure(rng, n_res, side)
    sts = [(base[0], base[1] + rng.normal(0.0, 1.0, base[1].shape), base[2]) for _ in range(4)]
    # a structure of another size with its own labels
    other = cg_structure(rng, 41, side)
    sts.append(other)
    rule = {"accept_same": False}
    lchd = lh.LoCoHD(CG_TYPES, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule(rule))
    ps = [prims(lh, *s) for s in sts]
    cent = [(i, i) for i in range(0, 3 * n_res, 3)]
    jobs = [(0, 1, cent), (0, 2, cent), (3, 0, cent[::2]), (1, 1, cent), (4, 0, [(0, 0), (3, 9), (120, 177)]), (2, 4, [])]
    got = lchd.from_primitives_batch(ps, jobs, 10.0)
    assert [len(g) for g in got] == [len(j[2]) for j in jobs]
    lo = oracle.LoCoHD(CG_TYPES, oracle.WeightFunction("uniform", [3.0, 10.0]), oracle.TagPairingRule(rule))
    po = [prims(oracle, *s) for s in sts]
    for (a, b, pairs), g in zip(jobs, got):
        if not pairs:
            assert g == []
            continue
        single = lchd.from_primitives(ps[a], ps[b], pairs, 10.0)
        assert g == single  # bitwise
        want = np.asarray(lo.from_primitives(po[a], po[b], pairs, 10.0))
        assert np.max(np.abs(np.asarray(g) - want)) < TIGHT
    assert np.max(np.abs(np.asarray(got[3]))) == 0.0  # a structure against itself
    with pytest.raises(lh.PanicException):
        lchd.from_primitives_batch(ps, [(0, 4, [(0, 500)])], 10.0)
    with pytest.raises(IndexError):
        lchd.from_primitives_batch(ps, [(0, 9, cent)], 10.0)


def test_small_pair_kernel_with_a_minority_of_large_pairs(lh, oracle, monkeypatch):
    """k_sweep_duo (two pairs per wavefront) takes the pairs with <= 240 merged events when they are the majority; the
    INDIRECT instantiation of k_sweep picks the larger ones out of the pair records.  A sparse cloud with one dense blob gives
    both kinds in one call; a second cloud (mostly dense) makes the small pairs a minority, where the plain kernel must do
    everything.  Checked against the oracle and against the same call with the small-pair kernel disabled."""
    rng = np.random.default_rng(21)
    cats = ["a", "b", "c", "d", "e", "f"]

    def cloud(n_sparse, n_dense):
        pts = np.concatenate([rng.uniform(0, 60, (n_sparse, 3)), rng.normal(30, 3.0, (n_dense, 3))])
        return [cats[i] for i in rng.integers(0, 6, len(pts))], pts

    for n_sparse, n_dense, expect_small_majority in ((3000, 260, True), (300, 900, False)):
        sa, xa = cloud(n_sparse, n_dense)
        sb, xb = cloud(n_sparse, n_dense)
        n = len(xa)
        anchors = [(i, int(j)) for i, j in zip(range(n), rng.permutation(n))] + [(5, 5), (n - 1, n - 1)]

        def run(mod):
            lchd = mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.25]))
            pa = [mod.PrimitiveAtom(t, "", c) for t, c in zip(sa, xa)]
            pb = [mod.PrimitiveAtom(t, "", c) for t, c in zip(sb, xb)]
            if mod is oracle:
                out, sizes = lchd.from_primitives(pa, pb, anchors, 8.0, return_env_sizes=True)
                return np.asarray(out), np.asarray(sizes)
            return np.asarray(lchd.from_primitives(pa, pb, anchors, 8.0)), None

        want, sizes = run(oracle)
        events = sizes.sum(axis=1) - 2 if sizes.ndim == 2 else None
        if events is not None:  # the intended mix of pair sizes
            small = np.mean(events <= 240)
            assert (small >= 0.5) == expect_small_majority and 0.02 < small < 0.98, small
        inline, _ = run(lh)  # a call this small: the one-launch sweep (records inline, one pair per wavefront)
        assert np.max(np.abs(inline - want)) < TIGHT
        monkeypatch.setenv("LCHD_NO_INLINE_META", "1")  # the regular pipeline: k_pair_meta, then the device picks the kernels
        got, _ = run(lh)
        assert np.max(np.abs(got - want)) < TIGHT
        monkeypatch.setenv("LCHD_NO_DUO", "1")
        plain, _ = run(lh)
        monkeypatch.delenv("LCHD_NO_DUO")
        monkeypatch.delenv("LCHD_NO_INLINE_META")
        assert np.max(np.abs(got - plain)) < 1e-13 and np.max(np.abs(inline - plain)) < 1e-13


def test_eight_bit_count_sweep_with_a_minority_of_large_environments(lh, oracle, monkeypatch):
    """More than 16 category slots: pairs whose environments both have <= 255 points go to the 8-bit-count k_sweep, the
    others to the INDIRECT 16-bit instantiation.  A cloud at protein-like density with one dense blob gives both kinds in one
    call (and environments of exactly 255 / 256 points sit on the boundary); a mostly dense cloud makes the large pairs the
    majority, where the plain kernel does everything.  Against the oracle and against the call with the 8-bit kernel disabled."""
    monkeypatch.setenv("LCHD_NO_INLINE_META", "1")
    rng = np.random.default_rng(27)
    for ncat in (17, 25, 32):
        cats = [f"k{i}" for i in range(ncat)]
        for n_sparse, n_dense, expect_small_majority in ((4000, 500, True), (300, 1200, False)):
            def cloud():
                pts = np.concatenate([rng.uniform(0, 43, (n_sparse, 3)), rng.normal(21, 2.6, (n_dense, 3))])
                return rng.integers(0, ncat, len(pts)).astype(np.int32), pts

            (ca, xa), (cb, xb) = cloud(), cloud()
            n = len(xa)
            pairs = np.stack([np.arange(n), rng.permutation(n)], 1).astype(np.int64)
            tag = np.zeros(n, dtype=np.int32)
            lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.2]))
            want, sizes = lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, 10.0, return_env_sizes=True)
            want, sizes = np.asarray(want), np.asarray(sizes)
            small = np.mean(sizes.max(axis=1) <= 255)
            assert (small >= 0.5) == expect_small_majority and 0.02 < small < 0.98, small
            pk = lambda x, c: lh.api._Packed(x, c, tag)
            got = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.2])).from_packed(pk(xa, ca), pk(xb, cb), pairs, 10.0)
            assert np.max(np.abs(got - want)) < TIGHT, (ncat, n_dense)
            monkeypatch.setenv("LCHD_NO_COUNT8", "1")
            plain = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.2])).from_packed(pk(xa, ca), pk(xb, cb), pairs, 10.0)
            monkeypatch.delenv("LCHD_NO_COUNT8")
            assert np.max(np.abs(got - plain)) < 1e-13


@pytest.mark.parametrize("ncat", [7, 11, 15])
def test_sweep_hint_follows_the_workload(lh, oracle, monkeypatch, ncat):
    """One context scores workloads whose pair sizes flip between 'mostly <= 240 events' (k_sweep_duo + indirect k_sweep),
    'environments <= 255 points' (the 8-bit-count k_sweep<8 / 12 / 16> + indirect) and 'larger' (plain k_sweep): the first
    pass launches every candidate and lets the device decide, later passes launch what the PREVIOUS pass's counts suggest.
    Any choice must give the oracle's scores for any input."""
    import torch
    from loco_hd_amd.device import DeviceSession

    monkeypatch.setenv("LCHD_NO_INLINE_META", "1")
    rng = np.random.default_rng(77 + ncat)
    cats = [f"c{i}" for i in range(ncat)]
    wf = ("hyper_exp", [1.0, 0.2])
    lchd = lh.LoCoHD(cats, lh.WeightFunction(*wf))
    lo = oracle.LoCoHD(cats, oracle.WeightFunction(*wf), n_of_threads=8)
    sess = DeviceSession(lchd)
    clouds = {}
    for name, density in (("sparse", 0.02), ("dense", 0.05), ("packed", 0.08)):  # ~84 / ~209 / ~335 points per environment at threshold 10
        n = 2500
        side = (n / density) ** (1 / 3)
        xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
        ca, cb = rng.integers(0, ncat, n).astype(np.int32), rng.integers(0, ncat, n).astype(np.int32)
        pairs = np.stack([rng.integers(0, n, 3000), rng.integers(0, n, 3000)], 1).astype(np.int64)
        tag = np.zeros(n, dtype=np.int32)
        want = np.asarray(lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, 10.0))
        clouds[name] = (sess.upload(xa, ca), sess.upload(xb, cb), torch.from_numpy(pairs).cuda(), want)
    order = ("sparse", "sparse", "dense", "dense", "sparse", "dense", "packed", "packed", "dense", "packed", "sparse", "packed", "dense", "dense")
    for name in order:  # every hint transition
        a, b, anchors, want = clouds[name]
        got = sess.from_primitives(a, b, anchors, 10.0).cpu().numpy()
        assert np.max(np.abs(got - want)) < TIGHT, name
    sess.close()


@pytest.mark.parametrize("ncat,density", [(7, 0.060), (7, 0.065), (8, 0.0575), (11, 0.0575), (12, 0.061), (15, 0.0575), (16, 0.060),
                                          (17, 0.0575), (20, 0.060), (23, 0.0575), (25, 0.061), (28, 0.0575), (29, 0.060), (32, 0.0575)])
def test_two_pairs_per_wavefront_eight_bit_sweep_around_its_limits(lh, oracle, monkeypatch, ncat, density):
    """Pairs whose environments both have <= 255 points AND that have <= 480 merged events are swept two per wavefront
    (k_sweep_duo<CMAX, 32, 480>, CMAX = 8 ... 32: one or two words of 4-bit chunk fields, one to four words of 8-bit counts); the others go to the INDIRECT 16-bit k_sweep.  Environments of ~212 ... ~244
    points put pairs on both sides of both limits in one call (at 0.065 atoms/A^3 the qualifying pairs are a minority: the plain
    k_sweep takes everything).  Against the oracle; the first pass of a context (no hint: every
    candidate kernel is launched, the device decides) and the second (the host launches what the first pass counted) must agree
    BIT FOR BIT -- both evaluate the same function of the pair list; and against the one-pair-per-wavefront 8-bit sweep."""
    import torch
    from loco_hd_amd.device import DeviceSession

    monkeypatch.setenv("LCHD_NO_INLINE_META", "1")
    rng = np.random.default_rng(int(density * 1e4) + ncat)
    n = 3000
    side = (n / density) ** (1 / 3)
    # (periodic images would be needed for every environment to reach the bulk size: the rim gives the smaller pairs)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, ncat, n).astype(np.int32), rng.integers(0, ncat, n).astype(np.int32)
    cats = [f"c{i}" for i in range(ncat)]
    centre = np.argsort(np.linalg.norm(xa - side / 2, axis=1))[: n // 2]  # anchors away from the rim: bulk-sized environments
    centre_b = np.argsort(np.linalg.norm(xb - side / 2, axis=1))[: n // 2]
    pairs = np.stack([rng.choice(centre, 6000), rng.choice(centre_b, 6000)], 1).astype(np.int64)
    tag = np.zeros(n, dtype=np.int32)
    wf = ("hyper_exp", [1.0, 0.2])
    lo = oracle.LoCoHD(cats, oracle.WeightFunction(*wf), n_of_threads=8)
    want, sizes = lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, 10.0, return_env_sizes=True)
    want, sizes = np.asarray(want), np.asarray(sizes)
    c8 = sizes.max(axis=1) <= 255
    team = c8 & (sizes.sum(axis=1) - 2 <= 480)
    assert 0.02 < np.mean(team) < 0.98 and np.mean(c8 & ~team) > 0.01 and np.mean(~c8) > 0.01, (np.mean(team), np.mean(c8))
    outs = {}
    for mode in ("team", "single"):
        if mode == "single":
            monkeypatch.setenv("LCHD_NO_C8_TEAM", "1")
        sess = DeviceSession(lh.LoCoHD(cats, lh.WeightFunction(*wf)))
        a, b, d_pairs = sess.upload(xa, ca), sess.upload(xb, cb), torch.from_numpy(pairs).cuda()
        first = sess.from_primitives(a, b, d_pairs, 10.0).cpu().numpy()
        second = sess.from_primitives(a, b, d_pairs, 10.0).cpu().numpy()
        third = sess.from_primitives(a, b, d_pairs, 10.0).cpu().numpy()
        sess.close()
        assert np.max(np.abs(first - want)) < TIGHT, (mode, np.mean(team))
        if mode == "team":
            assert np.array_equal(first, second) and np.array_equal(second, third)
        else:  # (the hook leaves the first pass to the plain 16-bit sweep, whose tiles cut long pairs differently: last-bit differences)
            assert np.max(np.abs(first - second)) < 1e-13 and np.array_equal(second, third)
        outs[mode] = first
    assert np.max(np.abs(outs["team"] - outs["single"])) < 1e-13


def test_leftover_list_of_hinted_passes_follows_a_changing_pair_list(lh, oracle, monkeypatch):
    """From the second pass of a context on, the record pass LISTS the pairs the team kernel's rule leaves over and the INDIRECT companion
    walks that list (sized from what the previous pass left) instead of scanning every record.  Lists whose leftovers grow from a
    dozen to hundreds and shrink again, on one session: every call against the oracle, and bit for bit against a fresh
    session's first (scanning) pass of the same list."""
    import torch
    from loco_hd_amd.device import DeviceSession

    monkeypatch.setenv("LCHD_NO_INLINE_META", "1")
    rng = np.random.default_rng(77)
    n, ncat = 4000, 10
    side = (n / 0.055) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, ncat, n).astype(np.int32), rng.integers(0, ncat, n).astype(np.int32)
    cats = [f"c{i}" for i in range(ncat)]
    tag = np.zeros(n, dtype=np.int32)
    wf = ("hyper_exp", [1.0, 0.1])
    ra, rb = np.linalg.norm(xa - side / 2, axis=1), np.linalg.norm(xb - side / 2, axis=1)
    rim_a, rim_b = np.argsort(ra)[n // 3:], np.argsort(rb)[n // 3:]        # smaller environments: the team rule takes (almost) all
    core_a, core_b = np.argsort(ra)[: n // 6], np.argsort(rb)[: n // 6]    # bulk-sized ones: many pairs beyond 480 merged events
    def mix(n_rim, n_core):
        return np.concatenate([np.stack([rng.choice(rim_a, n_rim), rng.choice(rim_b, n_rim)], 1),
                               np.stack([rng.choice(core_a, n_core), rng.choice(core_b, n_core)], 1)]).astype(np.int64)
    lists = [mix(9000, 20), mix(6000, 3000), mix(9000, 0), mix(7000, 1500), mix(9000, 5)]
    lo = oracle.LoCoHD(cats, oracle.WeightFunction(*wf), n_of_threads=8)
    sess = DeviceSession(lh.LoCoHD(cats, lh.WeightFunction(*wf)))
    a, b = sess.upload(xa, ca), sess.upload(xb, cb)
    left = []
    for k, pairs in enumerate(lists):
        want, sizes = lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, 10.0, return_env_sizes=True)
        want, sizes = np.asarray(want), np.asarray(sizes)
        left.append(int(np.sum(~((sizes.max(axis=1) <= 255) & (sizes.sum(axis=1) - 2 <= 480)))))
        got = sess.from_primitives(a, b, torch.from_numpy(pairs).cuda(), 10.0).cpu().numpy()
        assert np.max(np.abs(got - want)) < TIGHT, (k, left)
        fresh = DeviceSession(lh.LoCoHD(cats, lh.WeightFunction(*wf)))
        fa, fb = fresh.upload(xa, ca), fresh.upload(xb, cb)
        first = fresh.from_primitives(fa, fb, torch.from_numpy(pairs).cuda(), 10.0).cpu().numpy()
        fresh.close()
        assert np.array_equal(got, first), (k, left)
    sess.close()
    assert left[1] > 10 * max(left[0], 1) and left[2] * 10 < left[1] and left[3] > 100, left  # the list did grow and shrink (15, 403, 12, 193, 15)


def test_regular_batch_of_large_structures_uses_the_per_structure_cell_build(lh, oracle):
    """64 frames of an 11 000-atom structure: the one-workgroup-per-structure cell list at (almost) its LDS limit, against
    the oracle on a sample of frames and against the generic cell-list path."""
    import os

    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(31)
    n, nf = 11_000, 64
    side = (n / 0.05) ** (1 / 3)
    base = rng.uniform(0, side, (n, 3))
    cat = rng.integers(0, 6, n).astype(np.int32)
    cats = [f"c{i}" for i in range(6)]
    lchd = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.3]))
    frames = base[None] + rng.normal(0, 0.3, (nf, n, 3))
    la = np.arange(0, n, 37)
    outs = {}
    for mode in ("struct", "generic"):
        if mode == "generic":
            os.environ["LCHD_NO_STRUCT_CELLS"] = "1"
        try:
            sess = DeviceSession(lchd)
            ref = sess.upload(base, cat)
            outs[mode] = sess.score_trajectory(ref, frames, np.stack([la, la], 1), 7.0, chunk=nf)
            sess.close()
        finally:
            os.environ.pop("LCHD_NO_STRUCT_CELLS", None)
    assert np.array_equal(outs["struct"], outs["generic"])
    lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.3]))
    tag = np.zeros(n, dtype=np.int32)
    for f in (0, 63):
        want = np.asarray(lo.from_arrays(base, cat, tag, frames[f], cat, tag, np.stack([la, la], 1), 7.0))
        assert np.max(np.abs(outs["struct"][f] - want)) < TIGHT


@pytest.mark.parametrize("density", [0.026, 0.029, 0.034, 0.039, 0.045])
def test_pair_sizes_around_the_small_pair_threshold(lh, oracle, density):
    """Environment sizes spread around the 240-event tile of k_sweep_duo, so that the small pairs are sometimes the majority
    and sometimes not: whichever kernels the device picks, every score equals the oracle's."""
    rng = np.random.default_rng(int(density * 1e4))
    n = 2500
    side = (n / density) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, 9, n).astype(np.int32), rng.integers(0, 9, n).astype(np.int32)
    cats = [f"c{i}" for i in range(9)]
    pairs = np.stack([rng.integers(0, n, 20_000), rng.integers(0, n, 20_000)], 1).astype(np.int64)
    tag = np.zeros(n, dtype=np.int32)
    lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.15]))
    want, sizes = lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, 10.0, return_env_sizes=True)
    lchd = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.15]))
    got = lchd.from_packed(lh.api._Packed(xa, ca, tag), lh.api._Packed(xb, cb, tag), pairs, 10.0)
    assert np.max(np.abs(got - np.asarray(want))) < TIGHT
    small = np.mean(np.asarray(sizes).sum(axis=1) - 2 <= 240)
    assert 0.0 < small < 1.0  # the sweep really had both kinds of pairs to deal with (share of small pairs: 0.99 ... 0.27)


def test_same_object_on_both_sides_shares_environments(lh, oracle, monkeypatch):
    """Both sides ONE device object (all-vs-all inside a batch, a structure against itself): cell list and environments are
    built once for the anchors of both columns.  Bitwise equal to the two-sided build (LCHD_NO_SHARED_ENVS) on every prologue
    tier, and equal to the oracle."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(77)
    cats = [f"c{i}" for i in range(10)]

    def run(build, pairs_of, thr):
        out = []
        for share in (True, False):
            if share:
                monkeypatch.delenv("LCHD_NO_SHARED_ENVS", raising=False)
            else:
                monkeypatch.setenv("LCHD_NO_SHARED_ENVS", "1")
            lchd = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1]), lh.TagPairingRule({"accept_same": False}))
            sess = DeviceSession(lchd)
            obj = build(sess)
            pairs = torch.from_numpy(pairs_of()).cuda()
            for _ in range(2):  # (the second pass runs with the sweep hint of the first)
                sc = sess.from_primitives(obj, obj, pairs, thr).cpu().numpy()
            out.append(sc)
            sess.close()
        monkeypatch.delenv("LCHD_NO_SHARED_ENVS", raising=False)
        assert np.array_equal(out[0], out[1])
        return out[0]

    # general three-launch prologue: one structure of 6000 atoms against itself, random pairs (many atoms anchor in both columns)
    n = 6000
    side = (n / 0.05) ** (1 / 3)
    xyz, cat, tag = rng.uniform(0, side, (n, 3)), rng.integers(0, 10, n).astype(np.int32), rng.integers(0, 40, n).astype(np.int32)
    pairs = np.stack([rng.integers(0, n, 20_000), rng.integers(0, n, 20_000)], 1).astype(np.int64)
    got = run(lambda s: s.upload(xyz, cat, tag), lambda: pairs, 9.0)
    lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.1]), oracle.TagPairingRule({"accept_same": False}), n_of_threads=8)
    want = np.asarray(lo.from_arrays(xyz, cat, tag, xyz, cat, tag, pairs[:3000], 9.0))
    assert np.max(np.abs(got[:3000] - want)) < TIGHT
    assert np.all(got[pairs[:, 0] == pairs[:, 1]] == 0.0)
    # one-workgroup-per-structure cell lists: a batch of 12 structures of 900 atoms, all-vs-all
    m = 900
    structs = [(rng.uniform(0, 26.0, (m, 3)), rng.integers(0, 10, m).astype(np.int32), rng.integers(0, 300, m).astype(np.int32)) for _ in range(12)]
    la = np.arange(0, m, 3)
    offs_holder = {}

    def build_batch(s):
        b, offs = s.upload_batch(structs)
        offs_holder["o"] = offs
        return b

    def batch_pairs():
        o = offs_holder["o"]
        return np.concatenate([np.stack([o[i] + la, o[j] + la], 1) for i in range(12) for j in range(i, 12)]).astype(np.int64)

    got = run(build_batch, batch_pairs, 10.0)
    k = 0
    for i in range(12):
        for j in range(i, 12):
            if (i, j) in ((0, 0), (0, 1), (3, 9), (11, 11)):
                want = np.asarray(lo.from_arrays(*structs[i], *structs[j], np.stack([la, la], 1), 10.0))
                assert np.max(np.abs(got[k * len(la):(k + 1) * len(la)] - want)) < TIGHT, (i, j)
            k += 1
    # a small single structure against itself (the fused one-launch prologue is for two different objects)
    xs, cs, ts = structs[0]
    got = run(lambda s: s.upload(xs, cs, ts), lambda: np.stack([np.arange(m), np.arange(m)[::-1]], 1).astype(np.int64), 10.0)
    want = np.asarray(lo.from_arrays(xs, cs, ts, xs, cs, ts, np.stack([np.arange(m), np.arange(m)[::-1]], 1), 10.0))
    assert np.max(np.abs(got - want)) < TIGHT


@pytest.mark.parametrize("density,n_cat", [(0.05, 10), (0.023, 8), (0.05, 16), (0.023, 5)])
def test_category_weights_through_the_team_sweeps(lh, oracle, density, n_cat):
    """category_weights != 1 (src/locohd.rs:319-346, pmf.rs:47-63; python_codes/pisces/pisces_random_pairs.py:146-158) on pair lists
    long enough for the team sweeps: the first call of a context decides on the device, the later ones launch the weighted
    instantiations of k_sweep_duo (four pairs of <= 240 events or two 8-bit-count pairs per wavefront) and their companion."""
    rng = np.random.default_rng(int(density * 1000) + n_cat)
    n, n_pairs = 2500, 24000
    side = (n / density) ** (1 / 3)
    cats = [f"c{i}" for i in range(n_cat)]
    weights = [float(v) for v in rng.uniform(0.2, 3.0, n_cat)]
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    sa, sb = [cats[k] for k in rng.integers(0, n_cat, n)], [cats[k] for k in rng.integers(0, n_cat, n)]
    pairs = [(int(a), int(b)) for a, b in zip(rng.integers(0, n, n_pairs), rng.integers(0, n, n_pairs))]
    # a few long pairs among the short ones (the companion launch): a dense clump in both structures
    xa[:40] = xa[0] + rng.normal(0, 1.5, (40, 3))
    xb[:40] = xb[0] + rng.normal(0, 1.5, (40, 3))

    def build(mod):
        lchd = mod.LoCoHD(cats, mod.WeightFunction("hyper_exp", [1.0, 0.1]), category_weights=weights)
        return lchd, [mod.PrimitiveAtom(t, "", c) for t, c in zip(sa, xa)], [mod.PrimitiveAtom(t, "", c) for t, c in zip(sb, xb)]

    lo, pa, pb = build(oracle)
    want = np.asarray(lo.from_primitives(pa, pb, pairs, 10.0))
    lchd, pa, pb = build(lh)
    first = np.asarray(lchd.from_primitives(pa, pb, pairs, 10.0))
    second = np.asarray(lchd.from_primitives(pa, pb, pairs, 10.0))
    third = np.asarray(lchd.from_primitives(pa, pb, pairs, 10.0))
    assert np.max(np.abs(first - want)) < TIGHT
    assert np.max(np.abs(second - want)) < TIGHT
    assert np.array_equal(second, third)
    unit = np.asarray(lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1])).from_primitives(pa, pb, pairs, 10.0))
    assert np.max(np.abs(unit - second)) > 1e-4  # the weights do change the scores


@pytest.mark.parametrize("density,n_cat", [(0.05, 10), (0.023, 8), (0.05, 16), (0.023, 3)])
def test_kolmogorov_smirnov_through_the_team_sweeps(lh, oracle, density, n_cat, monkeypatch):
    """StatisticalDistance("Kolmogorov-Smirnov", []) (src/statistical_distances.rs:16-21: max_c |p_c - q_c|) with unit category weights on
    pair lists long enough for the team sweeps: integer counts, max_c |a_c N_b - b_c N_a| / (N_a N_b).  The KSM instantiations of
    k_sweep_duo against the oracle and against the generic one-pair-per-wavefront sweep (LCHD_FORCE_GENERIC)."""
    rng = np.random.default_rng(int(density * 1000) + n_cat + 77)
    n, n_pairs = 2500, 24000
    side = (n / density) ** (1 / 3)
    cats = [f"c{i}" for i in range(n_cat)]
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    sa, sb = [cats[k] for k in rng.integers(0, n_cat, n)], [cats[k] for k in rng.integers(0, n_cat, n)]
    pairs = [(int(a), int(b)) for a, b in zip(rng.integers(0, n, n_pairs), rng.integers(0, n, n_pairs))]
    xa[:40] = xa[0] + rng.normal(0, 1.5, (40, 3))  # a few long pairs among the short ones (the companion launch)
    xb[:40] = xb[0] + rng.normal(0, 1.5, (40, 3))

    def build(mod):
        lchd = mod.LoCoHD(cats, mod.WeightFunction("uniform", [3.0, 10.0]), statistical_distance=mod.StatisticalDistance("Kolmogorov-Smirnov", []))
        return lchd, [mod.PrimitiveAtom(t, "", c) for t, c in zip(sa, xa)], [mod.PrimitiveAtom(t, "", c) for t, c in zip(sb, xb)]

    lo, pa, pb = build(oracle)
    want = np.asarray(lo.from_primitives(pa, pb, pairs, 10.0))
    lchd, pa, pb = build(lh)
    first = np.asarray(lchd.from_primitives(pa, pb, pairs, 10.0))
    second = np.asarray(lchd.from_primitives(pa, pb, pairs, 10.0))
    third = np.asarray(lchd.from_primitives(pa, pb, pairs, 10.0))
    assert np.max(np.abs(first - want)) < TIGHT
    assert np.max(np.abs(second - want)) < TIGHT
    assert np.array_equal(second, third)
    monkeypatch.setenv("LCHD_FORCE_GENERIC", "1")
    generic = np.asarray(build(lh)[0].from_primitives(pa, pb, pairs, 10.0))
    assert np.max(np.abs(generic - second)) < TIGHT
    h2 = np.asarray(lh.LoCoHD(cats, lh.WeightFunction("uniform", [3.0, 10.0])).from_primitives(pa, pb, pairs, 10.0))
    assert np.max(np.abs(h2 - second)) > 1e-4  # (a different distance)


def test_trajectory_frames_with_more_than_255_categories(lh, oracle):
    """A frames buffer of a template with 300 categories (two-byte category ids travel with every frame; src/locohd.rs:312-316 takes
    any number of categories) against per-frame oracle calls."""
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(88)
    n, n_frames = 600, 9
    cats = [f"k{i}" for i in range(300)]
    seq = [cats[k] for k in rng.integers(0, 300, n)]
    xyz = rng.uniform(0.0, 24.0, (n, 3))
    tags = [f"A/{i // 3}" for i in range(n)]
    frames = xyz[None, :, :] + rng.normal(0.0, 0.4, (n_frames,) + xyz.shape)
    la = np.arange(0, n, 5)
    local_pairs = np.stack([la, la], 1)
    lchd = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.2]))
    interner = {}
    packed = lchd.pack(prims(lh, seq, xyz, tags), interner)
    sess = DeviceSession(lchd, interner=interner)
    ref = sess.upload(packed.xyz, packed.cat, packed.tag)
    got = sess.score_trajectory(ref, frames, local_pairs, 9.0, chunk=4)
    sess.close()
    lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.2]))
    ref_p = prims(oracle, seq, xyz, tags)
    for f in (0, 3, 4, 8):
        want = np.asarray(lo.from_primitives(ref_p, prims(oracle, seq, frames[f], tags), [(int(i), int(i)) for i in la], 9.0))
        assert np.max(np.abs(got[f] - want)) < TIGHT, f


@pytest.mark.parametrize("density,n_cat,n_wf", [(0.05, 10, 2), (0.023, 8, 3), (0.05, 16, 4), (0.023, 5, 5)])
def test_weight_function_dictionary_through_the_team_sweeps(lh, oracle, density, n_cat, n_wf, monkeypatch):
    """w_func as a dictionary + one key per anchor pair (src/locohd.rs:27-32,230-283).  Up to four functions: the environment store
    holds one set of F keys per function (k_env_key_sets) and every pair is swept from the set of its function by the same team
    kernels as a single-function configuration; five and more: distance keys, the CDF evaluated per event.  Against the oracle,
    against the distance-key path (LCHD_NO_KEY_SETS), and an index outside the dictionary."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(int(density * 1000) + n_cat + 31 * n_wf)
    n, n_pairs = 2500, 24000
    side = (n / density) ** (1 / 3)
    cats = [f"c{i}" for i in range(n_cat)]
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, n_cat, n).astype(np.int32), rng.integers(0, n_cat, n).astype(np.int32)
    pairs = np.stack([rng.integers(0, n, n_pairs), rng.integers(0, n, n_pairs)], 1).astype(np.int64)
    xa[:40] = xa[0] + rng.normal(0, 1.5, (40, 3))  # a few long pairs among the short ones (the companion launch)
    xb[:40] = xb[0] + rng.normal(0, 1.5, (40, 3))
    specs = [("hyper_exp", [1.0, 0.1]), ("uniform", [3.0, 10.0]), ("kumaraswamy", [2.0, 9.5, 1.7, 2.2]), ("dagum", [2.5, 6.0, 1.3]),
             ("hyper_exp", [0.3, 0.7, 0.25, 0.08])][:n_wf]
    wf_idx = rng.integers(0, n_wf, n_pairs).astype(np.int32)
    tz = np.zeros(n, dtype=np.int32)

    def build(mod):
        return mod.LoCoHD(cats, {f"w{k}": mod.WeightFunction(nm, prm) for k, (nm, prm) in enumerate(specs)})

    lo = build(oracle)
    want = np.asarray(lo.from_arrays(xa, ca, tz, xb, cb, tz, pairs, 10.0, *lo._wfs([f"w{k}" for k in wf_idx], n_pairs)))

    def run(idx):
        sess = DeviceSession(build(lh))
        a, b = sess.upload(xa, ca), sess.upload(xb, cb)
        anchors, wfi = torch.from_numpy(pairs).cuda(), torch.from_numpy(idx).cuda()
        outs = [sess.from_primitives(a, b, anchors, 10.0, wf_index=wfi).cpu().numpy() for _ in range(3)]
        sess.close()
        return outs

    outs = run(wf_idx)
    assert np.max(np.abs(outs[0] - want)) < TIGHT
    assert np.max(np.abs(outs[1] - want)) < TIGHT
    assert np.array_equal(outs[1], outs[2])
    monkeypatch.setenv("LCHD_NO_KEY_SETS", "1")
    plain = run(wf_idx)
    assert np.max(np.abs(plain[1] - outs[1])) < TIGHT
    monkeypatch.delenv("LCHD_NO_KEY_SETS")
    bad = wf_idx.copy()
    bad[777] = n_wf
    with pytest.raises(ValueError):
        run(bad)
    bad[777] = -1
    with pytest.raises(ValueError):
        run(bad)


def test_deterministic_switch_makes_a_pairs_score_independent_of_batch_and_history(lh, oracle):
    """lchd_ctx_set_deterministic (LoCoHD(..., deterministic=True), DeviceSession.set_deterministic): the reference is one code
    path (src/locohd.rs:61-226), so a pair's score cannot depend on the other pairs of the call.  With the switch on, the same
    probe pairs give the SAME BITS inside a list of small pairs, inside a list of large pairs, alone in a small call, after a
    different call history, reached through a second pass over overflowed environments, and from from_anchors-style single calls
    of the host API; with it off the calls still agree to 1e-13 and both match the oracle."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(2024)
    n_cat = 9
    cats = [f"c{i}" for i in range(n_cat)]
    n_sparse, n_dense, n_cluster = 30_000, 6_000, 1_500
    side = (n_sparse / 0.0025) ** (1 / 3)

    def cloud():
        v = rng.normal(0.0, 1.0, (n_cluster, 3))
        xyz = np.concatenate([rng.uniform(0.0, side, (n_sparse, 3)),                                   # ~10 points per environment
                              rng.uniform(0.0, 53.0, (n_dense, 3)) + np.array([-150.0, 0.0, 0.0]),    # ~170 points per environment
                              np.array([0.0, -150.0, 0.0]) + v / np.linalg.norm(v, axis=1)[:, None] * (4.0 * rng.uniform(0, 1, (n_cluster, 1)) ** (1 / 3))])
        return xyz, rng.integers(0, n_cat, len(xyz)).astype(np.int32), np.zeros(len(xyz), np.int32)

    xa, ca, ta = cloud()
    xb, cb, tb = cloud()
    n = 6000
    sparse_pairs = np.stack([rng.integers(0, n_sparse, n), rng.integers(0, n_sparse, n)], 1).astype(np.int64)
    dense_pairs = n_sparse + np.stack([rng.integers(0, n_dense, n), rng.integers(0, n_dense, n)], 1).astype(np.int64)
    probes = np.concatenate([sparse_pairs[:40], dense_pairs[:40],
                             np.stack([n_sparse + rng.integers(0, n_dense, 10), rng.integers(0, n_sparse, 10)], 1)]).astype(np.int64)
    cluster_pairs = np.stack([n_sparse + n_dense + rng.integers(0, n_cluster, 25), rng.integers(0, n_sparse, 25)], 1).astype(np.int64)
    want = np.asarray(oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.1])).from_arrays(xa, ca, ta, xb, cb, tb, probes, 10.0))

    def lists():
        k = len(probes)
        in_sparse = sparse_pairs.copy(); in_sparse[100:100 + k] = probes
        in_dense = dense_pairs.copy(); in_dense[2000:2000 + k] = probes
        with_cluster = sparse_pairs.copy(); with_cluster[300:300 + k] = probes; with_cluster[1000:1025] = cluster_pairs
        return [(in_sparse, 100), (in_dense, 2000), (probes.copy(), 0), (with_cluster, 300), (in_sparse, 100)]

    results = {}
    for det in (True, False):
        sess = DeviceSession(lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1]), deterministic=det))
        a, b = sess.upload(xa, ca, ta), sess.upload(xb, cb, tb)
        got = []
        for pairs, at in lists():
            out = sess.from_primitives(a, b, torch.from_numpy(pairs).cuda(), 10.0).cpu().numpy()
            got.append(out[at:at + len(probes)])
        if det:
            assert sess.pass_counts()["subset_passes"] >= 1  # (the cluster's pairs went through a second pass)
        sess.close()
        results[det] = got
        for g in got:
            assert np.max(np.abs(g - want)) < 1e-11
    for g in results[True][1:]:
        assert np.array_equal(g, results[True][0])  # bit for bit, whatever surrounded the probes
    for g in results[False]:
        assert np.max(np.abs(g - results[True][0])) < 1e-13
    # the host API (lists of PrimitiveAtom in, list of floats out) on a sub-structure: two different call shapes, same bits
    sub = np.arange(n_sparse, n_sparse + 1500)
    pa = [lh.PrimitiveAtom(cats[c], "", x) for c, x in zip(ca[sub], xa[sub])]
    pb = [lh.PrimitiveAtom(cats[c], "", x) for c, x in zip(cb[sub], xb[sub])]
    det = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1]), deterministic=True)
    all_pairs = [(i, i) for i in range(1500)]
    whole = np.asarray(det.from_primitives(pa, pb, all_pairs * 4, 10.0))[:1500]  # 6000 pairs: the regular pass
    few = np.asarray(det.from_primitives(pa, pb, all_pairs[:64], 10.0))          # 64 pairs: would be the one-launch sweep by default
    assert np.array_equal(whole[:64], few)


def test_deterministic_switch_gives_the_same_bits_every_run_on_lattice_inputs(lh, oracle):
    """The reference sorts an environment with a stable sort (utils.rs:25-39): one input, one order, one score.  The environment
    kernels place points of EQUAL distance in the order their atomics happened to complete -- another order in another run, and the
    O(1) Hellinger update then rounds differently (a few 1e-16).  Under the determinism switch the categories inside every run of
    equal keys are sorted (k_env_canon): lattice structures -- dozens of exact ties per environment, between points of different
    categories -- scored 20 times through from_primitives (cell lists with global atomics) and through from_coords (row sorts with
    LDS atomics over 16 wavefronts) give the same bits every time, and match the oracle."""
    import itertools

    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(606)
    n_cat = 7
    cats = [f"c{i}" for i in range(n_cat)]
    grid = np.array(list(itertools.product(range(16), repeat=3)), dtype=float) * 1.5
    xa = grid[rng.permutation(len(grid))]
    xb = grid[rng.permutation(len(grid))] + np.array([0.75, 0.0, 0.0])
    ca, cb = rng.integers(0, n_cat, len(xa)).astype(np.int32), rng.integers(0, n_cat, len(xb)).astype(np.int32)
    tag = np.zeros(len(xa), dtype=np.int32)
    pairs = np.stack([rng.integers(0, len(xa), 5000), rng.integers(0, len(xb), 5000)], 1).astype(np.int64)
    for wf in (("hyper_exp", [1.0, 0.2]), ("uniform", [1.0, 5.0])):  # (the uniform CDF adds long runs of equal F below x_min)
        want = np.asarray(oracle.LoCoHD(cats, oracle.WeightFunction(*wf), n_of_threads=8).from_arrays(xa, ca, tag, xb, cb, tag, pairs[:600], 6.0))
        sess = DeviceSession(lh.LoCoHD(cats, lh.WeightFunction(*wf), deterministic=True))
        a, b, d_pairs = sess.upload(xa, ca, tag), sess.upload(xb, cb, tag), torch.from_numpy(pairs).cuda()
        first = sess.from_primitives(a, b, d_pairs, 6.0).cpu().numpy()
        assert np.max(np.abs(first[:600] - want)) < 1e-11
        for _ in range(19):
            assert np.array_equal(sess.from_primitives(a, b, d_pairs, 6.0).cpu().numpy(), first)
        sess.close()
        other = DeviceSession(lh.LoCoHD(cats, lh.WeightFunction(*wf), deterministic=True))  # another context, another workspace: the same bits
        a, b = other.upload(xa, ca, tag), other.upload(xb, cb, tag)
        assert np.array_equal(other.from_primitives(a, b, d_pairs, 6.0).cpu().numpy(), first)
        other.close()
    # dense rows: 12^3 lattice points, every row of the distance matrix is full of ties
    g2 = np.array(list(itertools.product(range(12), repeat=3)), dtype=float)
    ya, yb = g2[rng.permutation(len(g2))], g2[rng.permutation(len(g2))]
    sa, sb = rng.choice(cats, len(ya)).tolist(), rng.choice(cats, len(yb)).tolist()
    det = lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.3]), deterministic=True)
    first = np.asarray(det.from_coords(sa, sb, ya, yb))
    want = np.asarray(oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.3])).from_coords(sa[:40], sb[:40], ya[:40], yb[:40]))
    sub = np.asarray(lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.3]), deterministic=True).from_coords(sa[:40], sb[:40], ya[:40], yb[:40]))
    assert np.max(np.abs(sub - want)) < 1e-11
    for _ in range(19):
        assert np.array_equal(np.asarray(det.from_coords(sa, sb, ya, yb)), first)


@pytest.mark.parametrize("n_cat,sd", [(5, None), (8, None), (12, None), (16, None), (11, ("Kolmogorov-Smirnov", []))])
def test_prefix_count_rows_give_the_same_bits_as_the_per_tile_histogram(lh, oracle, monkeypatch, n_cat, sd):
    """Configurations of at most 16 categories: k_env_group writes a prefix-count row per four points next to
    every environment (EnvStore::pre) and the team sweeps read a chunk's start counts from them -- the row below the chunk's start plus
    the one-hot fields of up to three category bytes (lchd_team_tile.h, PRE) -- instead of a histogram + scan per tile: integer counts
    either way, so the scores are bitwise those of LCHD_PRE_ROWS=-1; both follow the oracle (pmf.rs:47-63 counts the same points).
    One object on both sides, a weight-function dictionary and a tag rule ride along.  (Bitwise on coordinates without exact distance
    ties; a second structure on a lattice is compared with the oracle only: the order of TIED points of different categories inside an
    environment follows the cell list's arrival order, which atomics decide per run -- zero-width intervals, but the running
    Bhattacharyya sum rounds differently: a handful of pairs move by ~1e-16 from run to run, with or without rows.)"""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(40 + n_cat)
    cats = [f"c{i}" for i in range(n_cat)]
    n = 2600
    side = (n / 0.03) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    xl = np.round(rng.uniform(0, side, (n, 3)) * 2) / 2  # a lattice: many exact distance ties
    ca, cb = rng.integers(0, n_cat, n).astype(np.int32), rng.integers(0, n_cat, n).astype(np.int32)
    tag = (np.arange(n) // 3).astype(np.int32)
    pairs = np.stack([rng.integers(0, n, 9000), rng.integers(0, n, 9000)], 1).astype(np.int64)  # every anchor in several pairs
    wfi = (np.arange(9000) % 2).astype(np.int32)

    def build(mod, dictionary):
        wf = mod.WeightFunction("hyper_exp", [1.0, 0.12])
        if dictionary:
            wf = {"a": wf, "b": mod.WeightFunction("uniform", [2.0, 9.5])}
        kw = {} if sd is None else {"statistical_distance": mod.StatisticalDistance(*sd)}
        return mod.LoCoHD(cats, wf, mod.TagPairingRule({"accept_same": False}), **kw)

    for dictionary in (False, True):
        lo = build(oracle, dictionary)
        extra = lo._wfs(["ab"[k] for k in wfi], len(pairs)) if dictionary else ()
        want_ab = np.asarray(lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, 9.0, *extra))
        want_aa = np.asarray(lo.from_arrays(xa, ca, tag, xa, ca, tag, pairs, 9.0, *extra))
        want_lb = np.asarray(lo.from_arrays(xl, ca, tag, xb, cb, tag, pairs, 9.0, *extra))
        got = {}
        for hook in ("-1", "0"):
            monkeypatch.setenv("LCHD_PRE_ROWS", hook)
            sess = DeviceSession(build(lh, dictionary))
            monkeypatch.delenv("LCHD_PRE_ROWS")
            a, b = sess.upload(xa, ca, tag), sess.upload(xb, cb, tag)
            d_pairs = torch.from_numpy(pairs).cuda()
            d_wfi = torch.from_numpy(wfi).cuda() if dictionary else None
            outs = []
            for _ in range(3):  # first pass: the device picks the sweeps; later passes: the hinted launch set
                outs.append((sess.from_primitives(a, b, d_pairs, 9.0, wf_index=d_wfi).cpu().numpy(),
                             sess.from_primitives(a, a, d_pairs, 9.0, wf_index=d_wfi).cpu().numpy()))
            lat = sess.upload(xl, ca, tag)
            for _ in range(2):
                assert np.max(np.abs(sess.from_primitives(lat, b, d_pairs, 9.0, wf_index=d_wfi).cpu().numpy() - want_lb)) < 1e-11
            sess.close()
            got[hook] = outs
            for ab, aa in outs:
                assert np.max(np.abs(ab - want_ab)) < 1e-11 and np.max(np.abs(aa - want_aa)) < 1e-11
        for (ab0, aa0), (ab1, aa1) in zip(got["-1"], got["0"]):
            assert np.array_equal(ab0, ab1) and np.array_equal(aa0, aa1)


@pytest.mark.parametrize("n_cat", [6, 13])
def test_prefix_count_rows_on_environments_of_a_few_points(lh, oracle, monkeypatch, n_cat):
    """Rows exist for every FOURTH point of an environment (EnvStore::pre, kPreStep): a chunk that starts at point i reads row i / 4 and
    adds the i % 4 category bytes behind it (lchd_team_tile.h).  Sparse clouds and several thresholds give environments of 1, 2, ... ~15
    points -- every residue of the row step on both sides, anchors alone, rows that are an environment's last point -- swept with rows
    (forced for a call of any size) and without: the same bits, both at the oracle."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(700 + n_cat)
    cats = [f"c{i}" for i in range(n_cat)]
    n = 900
    side = (n / 0.004) ** (1 / 3)
    xa, xb = rng.uniform(0, side, (n, 3)), rng.uniform(0, side, (n, 3))
    ca, cb = rng.integers(0, n_cat, n).astype(np.int32), rng.integers(0, n_cat, n).astype(np.int32)
    tag = np.zeros(n, dtype=np.int32)
    pairs = np.stack([rng.integers(0, n, 6000), rng.integers(0, n, 6000)], 1).astype(np.int64)
    lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.2]))
    sizes = set()
    for thr in (4.0, 7.5, 9.0):
        want = np.asarray(lo.from_arrays(xa, ca, tag, xb, cb, tag, pairs, thr))
        got = {}
        for hook in ("-1", "1"):
            monkeypatch.setenv("LCHD_PRE_ROWS", hook)
            sess = DeviceSession(lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.2])))
            monkeypatch.delenv("LCHD_PRE_ROWS")
            a, b = sess.upload(xa, ca, tag), sess.upload(xb, cb, tag)
            d_pairs = torch.from_numpy(pairs).cuda()
            outs = [sess.from_primitives(a, b, d_pairs, thr).cpu().numpy() for _ in range(3)]  # (the third pass runs the hinted launch set)
            sess.close()
            for o in outs:
                assert np.max(np.abs(o - want)) < 1e-11
            got[hook] = outs
        for o0, o1 in zip(got["-1"], got["1"]):
            assert np.array_equal(o0, o1)
        d = np.sqrt(((xa[pairs[:200, 0], None, :] - xa[None, :, :]) ** 2).sum(-1))
        sizes |= set((d < thr).sum(1).tolist())
    assert {1, 2, 3, 4, 5, 6, 7, 8, 9} <= sizes  # (environment sizes the thresholds produced, anchor included)
