"""The thresholded environment build has two kernels and a fall-back chain between them (loco_hd_amd/csrc/lchd_env_group.hip,
lchd_env_cells.hip): k_env_group (several environments per wavefront, 320- and 512-point instantiations, half-threshold grid)
and k_env_cells / k_env_collect (one environment per workgroup, growing capacity).  Every link of the chain is forced here onto
inputs the CPU oracle can follow (reference: env_from_idx, /root/reference/src/locohd.rs:514-542)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT = 1e-11
CATS = ["A", "B", "C", "D", "E"]


@pytest.fixture(scope="module")
def lh():
    import loco_hd_amd

    return loco_hd_amd


def prims(mod, seq, xyz, tags=None):
    tags = [""] * len(seq) if tags is None else tags
    return [mod.PrimitiveAtom(s, t, c) for s, t, c in zip(seq, tags, xyz)]


def score(mod, seq_a, xyz_a, seq_b, xyz_b, anchors, thr, wf=("hyper_exp", [1.0, 0.1]), rule=None, tags_a=None, tags_b=None, repeat=1):
    lchd = mod.LoCoHD(CATS, mod.WeightFunction(*wf), *([] if rule is None else [mod.TagPairingRule(rule)]))
    pa, pb = prims(mod, seq_a, xyz_a, tags_a), prims(mod, seq_b, xyz_b, tags_b)
    return [np.asarray(lchd.from_primitives(pa, pb, anchors, thr)) for _ in range(repeat)]


def test_grouped_and_single_environment_kernels_agree(lh, oracle, monkeypatch):
    """Protein-like density: the first call of an object runs the 512-point instantiation with 2 anchors per wavefront, the
    later ones the instantiation / anchors per wavefront picked from the first call's largest environment; all of them, every
    forced number of anchors per wavefront and the one-environment-per-workgroup kernel give bitwise identical scores."""
    rng = np.random.default_rng(71)
    for n, side, thr in ((900, 26.0, 10.0), (700, 31.0, 10.0), (400, 12.0, 4.0)):
        sa, xa = rng.choice(CATS, n).tolist(), rng.uniform(0, side, (n, 3))
        sb, xb = rng.choice(CATS, n - 50).tolist(), rng.uniform(0, side, (n - 50, 3))
        anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, n, 600), rng.integers(0, n - 50, 600))]
        want = score(oracle, sa, xa, sb, xb, anchors, thr)[0]
        got = score(lh, sa, xa, sb, xb, anchors, thr, repeat=3)
        for g in got:
            assert np.max(np.abs(g - want)) < TIGHT
            assert np.array_equal(g, got[0])
        for hook in ({"LCHD_NO_ENV_GROUP": "1"}, {"LCHD_ENV_APW": "1"}, {"LCHD_ENV_APW": "3"}, {"LCHD_ENV_APW": "7"}, {"LCHD_ENV_APW": "40"}):
            for k, v in hook.items():
                monkeypatch.setenv(k, v)
            alt = score(lh, sa, xa, sb, xb, anchors, thr, repeat=2)
            for k in hook:
                monkeypatch.delenv(k)
            for a in alt:
                assert np.array_equal(a, got[0]), hook


def test_grouped_kernel_with_tag_rules(lh, oracle):
    """Both tag-rule instantiations of the grouped kernel (one comparison / pair-list search), residues of three atoms."""
    rng = np.random.default_rng(72)
    n = 600
    sa, xa = rng.choice(CATS, n).tolist(), rng.uniform(0, 24.0, (n, 3))
    sb, xb = rng.choice(CATS, n).tolist(), rng.uniform(0, 24.0, (n, 3))
    tags = [f"A/{i // 3}-RES" for i in range(n)]
    names = sorted(set(tags))
    pairs = [(names[int(i)], names[int(j)]) for i, j in zip(rng.integers(0, len(names), 4000), rng.integers(0, len(names), 4000))]
    anchors = [(i, i) for i in range(0, n, 2)]
    for rule in ({"accept_same": False}, {"accept_same": True},
                 {"tag_pairs": set(pairs), "accepted_pairs": True, "ordered": False},
                 {"tag_pairs": set(pairs), "accepted_pairs": False, "ordered": True}):
        want = score(oracle, sa, xa, sb, xb, anchors, 10.0, wf=("uniform", [3.0, 10.0]), rule=rule, tags_a=tags, tags_b=tags)[0]
        for g in score(lh, sa, xa, sb, xb, anchors, 10.0, wf=("uniform", [3.0, 10.0]), rule=rule, tags_a=tags, tags_b=tags, repeat=2):
            assert np.max(np.abs(g - want)) < TIGHT, rule


def test_fallback_chain_of_the_environment_kernels(lh, oracle):
    """Environments of 300 .. 320 points (the small instantiation's limit), 321 .. 512 (small -> regular instantiation),
    beyond 512 (-> k_env_cells with a larger capacity) and a neighbourhood with more candidate groups than the grouped
    kernel's table holds although the environment itself is small (-> k_env_cells).  ONE LoCoHD object scores a sparse
    structure in front of every case, so that each case starts from the small instantiation (a context picks its kernel
    from what its previous pass saw)."""
    rng = np.random.default_rng(73)
    filler = rng.uniform(40.0, 70.0, (300, 3))  # far away from the probe anchor
    xs = rng.uniform(0.0, 30.0, (500, 3))       # the sparse structure (~40 points per environment); also side B of every case
    ss = rng.choice(CATS, 500).tolist()
    sparse_anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, 500, 100), rng.integers(0, 500, 100))]
    mods = {}
    for mod in (lh, oracle):
        mods[mod] = (mod.LoCoHD(CATS, mod.WeightFunction("hyper_exp", [1.0, 0.1])), prims(mod, ss, xs))
    for n_in, n_shell in ((310, 0), (330, 0), (500, 0), (520, 0), (700, 0), (120, 2600)):
        centre = np.array([20.0, 20.0, 20.0])
        v = rng.normal(size=(n_in, 3))
        inner = centre + v / np.linalg.norm(v, axis=1)[:, None] * rng.uniform(0.2, 7.9, (n_in, 1))
        s_ = rng.normal(size=(n_shell, 3))
        shell = centre + s_ / np.maximum(np.linalg.norm(s_, axis=1)[:, None], 1e-9) * rng.uniform(8.05, 9.5, (n_shell, 1))  # just outside thr = 8
        xa = np.concatenate([centre[None], inner, shell, filler])
        sa = rng.choice(CATS, len(xa)).tolist()
        anchors = [(0, int(j)) for j in rng.integers(0, 500, 40)] + [(int(i), int(j)) for i, j in zip(rng.integers(0, len(xa), 200), rng.integers(0, 500, 200))]
        res = {}
        for mod, (lchd, pb) in mods.items():
            pa = prims(mod, sa, xa)
            out = []
            for _ in range(2):  # (the second round starts from the capacity / instantiation the first one ended with)
                out.append(np.asarray(lchd.from_primitives(pb, pb, sparse_anchors, 8.0)))
                out.append(np.asarray(lchd.from_primitives(pb, pb, sparse_anchors, 8.0)))
                out.append(np.asarray(lchd.from_primitives(pa, pb, anchors, 8.0)))
            res[mod] = out
        for g, w in zip(res[lh], res[oracle]):
            assert np.max(np.abs(g - w)) < TIGHT, (n_in, n_shell)


def test_grouped_kernel_on_lattices_and_flat_structures(lh, oracle):
    """Exact distance ties (a cubic lattice: whole shells of equal keys share one sort bucket), a planar and a collinear
    structure (most of the 5 x 5 x 5 neighbourhood is outside the grid), a structure smaller than one cell."""
    rng = np.random.default_rng(74)
    g = np.arange(9, dtype=np.float64) * 1.5
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    planar = np.concatenate([rng.uniform(0, 40, (500, 2)), np.zeros((500, 1))], 1)
    line = np.stack([np.linspace(0, 90, 400), np.full(400, 3.0), np.full(400, -2.0)], 1)
    tiny = rng.uniform(0, 1.5, (60, 3))
    for xa in (lattice, planar, line, tiny):
        n = len(xa)
        xb = xa + rng.normal(0, 0.3, xa.shape) if xa is not lattice else xa.copy()
        sa, sb = rng.choice(CATS, n).tolist(), rng.choice(CATS, n).tolist()
        anchors = [(int(i), int(j)) for i, j in zip(rng.integers(0, n, 300), rng.integers(0, n, 300))]
        for thr, wf in ((6.0, ("uniform", [1.0, 6.0])), (4.0, ("dagum", [2.5, 3.0, 1.2]))):
            want = score(oracle, sa, xa, sb, xb, anchors, thr, wf=wf)[0]
            for got in score(lh, sa, xa, sb, xb, anchors, thr, wf=wf, repeat=2):
                assert np.max(np.abs(got - want)) < TIGHT


def test_packed_list_cache_sees_every_change(lh, oracle):
    """LoCoHD keeps the packed form of the PrimitiveAtom lists it was given (the reference's callers pass the same structure
    again and again, python_codes/casp14/casp14_extend_with_locohd.py:72-79).  Every way of changing what a list means between
    two calls must be seen: an atom's setter, an item replaced, the list grown, another anchor list, another tag."""
    rng = np.random.default_rng(75)
    n = 400
    sa, xa = rng.choice(CATS, n).tolist(), rng.uniform(0, 20.0, (n, 3))
    sb, xb = rng.choice(CATS, n).tolist(), rng.uniform(0, 20.0, (n, 3))
    tags = [f"r{i // 4}" for i in range(n)]
    anchors = [(i, (7 * i) % n) for i in range(0, n, 2)]
    state = {}
    for mod in (lh, oracle):
        state[mod] = dict(lchd=mod.LoCoHD(CATS, mod.WeightFunction("uniform", [2.0, 9.0]), mod.TagPairingRule({"accept_same": False})),
                          pa=prims(mod, sa, xa, tags), pb=prims(mod, sb, xb, tags), anchors=list(anchors))

    def both():
        got = {mod: np.asarray(st["lchd"].from_primitives(st["pa"], st["pb"], st["anchors"], 9.0)) for mod, st in state.items()}
        assert np.max(np.abs(got[lh] - got[oracle])) < TIGHT
        return got[lh]

    first = both()
    assert np.array_equal(both(), first) and np.array_equal(both(), first)  # served from the cache, bitwise the same
    for mod, st in state.items():  # setters of atoms already in the lists
        st["pa"][10].coordinates = [1.0, 2.0, 3.0]
        st["pb"][11].primitive_type = "E" if sb[11] != "E" else "A"
        st["pa"][12].tag = "somewhere else"
    second = both()
    assert not np.array_equal(second, first)
    for mod, st in state.items():  # an item replaced by a new atom (same length)
        st["pb"][20] = mod.PrimitiveAtom("C", "r5", [3.0, 3.0, 3.0])
    third = both()
    assert not np.array_equal(third, second)
    for mod, st in state.items():  # the list grows; an anchor pair is replaced (same list object)
        st["pa"].append(mod.PrimitiveAtom("D", "new", [4.0, 4.0, 4.0]))
        st["anchors"][0] = (n, 3)
    fourth = both()
    assert fourth[0] != third[0]
    # a second LoCoHD object on the same lists shares nothing with the first one's cache
    other = lh.LoCoHD(CATS, lh.WeightFunction("uniform", [2.0, 9.0]), lh.TagPairingRule({"accept_same": True}))
    same_tags = np.asarray(other.from_primitives(state[lh]["pa"], state[lh]["pb"], state[lh]["anchors"], 9.0))
    want = np.asarray(oracle.LoCoHD(CATS, oracle.WeightFunction("uniform", [2.0, 9.0]), oracle.TagPairingRule({"accept_same": True}))
                      .from_primitives(state[oracle]["pa"], state[oracle]["pb"], state[oracle]["anchors"], 9.0))
    assert np.max(np.abs(same_tags - want)) < TIGHT
    # one list on both sides, tuples instead of lists, objects that merely look like PrimitiveAtoms (never cached)
    class Duck:
        def __init__(self, a):
            self.primitive_type, self.tag, self.coordinates = a.primitive_type, a.tag, a.coordinates
    lchd = state[lh]["lchd"]
    pa = state[lh]["pa"]
    self_scores = np.asarray(lchd.from_primitives(pa, pa, [(i, i) for i in range(50)], 9.0))
    assert np.all(self_scores == 0.0)
    ducks = [Duck(a) for a in pa]
    d1 = np.asarray(lchd.from_primitives(tuple(pa), ducks, [(i, i) for i in range(50)], 9.0))
    assert np.all(d1 == 0.0)
    ducks[3].coordinates = [9.0, 9.0, 9.0]
    d2 = np.asarray(lchd.from_primitives(tuple(pa), ducks, [(i, i) for i in range(50)], 9.0))
    assert d2[3] > 0.0
    # NEW list objects on every call (no cache entry can serve them): the atoms carry the ids of their strings (interned when an atom
    # is constructed or changed, primitive_atom.rs:4-25 has get + set), so a setter between two calls must show in the next call
    for mod, st in state.items():
        st["pa2"], st["pb2"] = list(st["pa"]), list(st["pb"])

    def fresh():
        got = {mod: np.asarray(st["lchd"].from_primitives(list(st["pa2"]), list(st["pb2"]), list(st["anchors"]), 9.0)) for mod, st in state.items()}
        assert np.max(np.abs(got[lh] - got[oracle])) < TIGHT
        return got[lh]

    f1 = fresh()
    assert np.array_equal(fresh(), f1)
    for mod, st in state.items():
        st["pa2"][30].tag = "a tag nobody else has"          # leaves its residue: it is no longer filtered out of its neighbours' environments
        st["pb2"][31].primitive_type = "A" if st["pb2"][31].primitive_type != "A" else "B"
        st["pb2"][32].coordinates = (0.5, 0.25, 0.125)
    f2 = fresh()
    assert not np.array_equal(f2, f1)
    for mod, st in state.items():  # a type that is not in this instance's category map: "Category not found!" (pmf.rs:38-42)
        st["pa2"][33].primitive_type = "not a category"
    for mod, st in state.items():
        with pytest.raises(ValueError):
            st["lchd"].from_primitives(list(st["pa2"]), list(st["pb2"]), list(st["anchors"]), 9.0)
    for mod, st in state.items():
        st["pa2"][33].primitive_type = "C"
    fresh()  # (the scores follow the oracle again: asserted inside)


def test_category_ids_beyond_the_map_raise_under_a_narrow_configuration():
    """A device-resident structure whose category array carries ids of 255 and more (raw C-ABI callers, DeviceSession.upload):
    under a configuration of at most 255 categories such an id is outside the map -- pmf.rs:38-42 raises for every environment
    that holds the atom -- and must not be scored by its low byte (id 256 is not category 0, id 300 not category 44)."""
    import torch
    import loco_hd_amd as lh
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(91)
    n = 400
    xyz = rng.uniform(0.0, 20.0, (n, 3))
    cat = rng.integers(0, 10, n).astype(np.int32)
    lchd = lh.LoCoHD([f"c{i}" for i in range(10)], lh.WeightFunction("uniform", [3.0, 10.0]))
    sess = DeviceSession(lchd)
    good = sess.upload(xyz, cat)
    anchors = torch.tensor([(i, i) for i in range(n)], dtype=torch.int64, device="cuda")
    want = sess.from_primitives(good, good, anchors, 8.0).cpu().numpy()
    assert np.max(np.abs(want)) == 0.0
    for bad_id in (256, 300, 255, 70000):
        c2 = cat.copy()
        c2[17] = bad_id
        bad = sess.upload(xyz, c2)
        with pytest.raises(ValueError):
            sess.from_primitives(good, bad, anchors, 8.0)
        with pytest.raises(ValueError):
            sess.from_coords(good, bad)
    again = sess.from_primitives(good, good, anchors, 8.0).cpu().numpy()  # the context still works
    assert np.array_equal(again, want)
    sess.close()


def test_small_calls_are_one_pass_in_the_steady_state(lh):
    """A from_primitives call is ONE pass once the context has seen the shape: no capacity retry, no repeat for the sweep launch set
    (the reference's per-call work, src/locohd.rs:479-567, has no retries at all).  Host-pointer calls of 1000 atoms / 334 pairs and
    3000 atoms / 1000 pairs, alternating."""
    from loco_hd_amd import _native as N

    types = ["Cent", "AmideC", "OH", "Pos", "Neg", "Aro", "Ali", "Sulf"]
    lchd = lh.LoCoHD(types, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule({"accept_same": False}))
    calls = []
    for nn in (1000, 3000):
        rs = np.random.default_rng(nn)
        sd = (nn / 0.023) ** (1 / 3)
        xa, xb = rs.uniform(0, sd, (nn, 3)), rs.uniform(0, sd, (nn, 3))
        ct = rs.integers(0, 8, nn).astype(np.int32)
        tg = (np.arange(nn) // 3).astype(np.int32)
        anchors = np.stack([np.arange(0, nn, 3), np.arange(0, nn, 3)], 1)
        calls.append((lh.api._Packed(xa, ct, tg), lh.api._Packed(xb, ct, tg), anchors))
    for pa, pb, an in calls * 3:  # warm-up: both shapes, capacity and launch-set hints settled
        lchd.from_packed(pa, pb, an, 10.0)
    ctx = lchd._context()
    for pa, pb, an in calls * 10:
        before = N.lib().lchd_ctx_pass_count(ctx)
        first = np.asarray(lchd.from_packed(pa, pb, an, 10.0))
        assert N.lib().lchd_ctx_pass_count(ctx) - before == 1
        assert np.array_equal(first, np.asarray(lchd.from_packed(pa, pb, an, 10.0)))


def _cluster_in_sparse_cloud(seed, n_sparse, n_cluster, n_cat):
    """A sparse cloud (environments of ~10 points at 10 A) with ONE dense cluster (every cluster point sees the whole cluster)."""
    rng = np.random.default_rng(seed)
    side = (n_sparse / 0.0025) ** (1 / 3)
    xyz = rng.uniform(0.0, side, (n_sparse + n_cluster, 3))
    centre = xyz[0] + 40.0
    v = rng.normal(0.0, 1.0, (n_cluster, 3))
    xyz[n_sparse:] = centre + v / np.linalg.norm(v, axis=1)[:, None] * (4.5 * rng.uniform(0.0, 1.0, (n_cluster, 1)) ** (1 / 3))
    cat = rng.integers(0, n_cat, len(xyz)).astype(np.int32)
    tag = rng.integers(0, 50, len(xyz)).astype(np.int32)
    return xyz, cat, tag


@pytest.mark.parametrize("variant", ["two_structures", "same_object", "weight_function_dictionary", "tag_rule", "one_environment_per_workgroup",
                                     "cluster_of_20000"])
def test_a_dense_cluster_in_a_sparse_cloud_rescoring_only_its_pairs(lh, oracle, variant, monkeypatch):
    """Environments have no capacity in the reference (src/locohd.rs:514-542: a Vec per anchor, utils.rs:25-39 sorts whatever
    it holds); here they live in fixed-stride slots of 512 points.  One 3 000-point cluster inside a sparse 60 000-point cloud: the
    cluster's anchors overflow their slots, and ONLY the pairs that touch them are scored again (a second pass with 4 096-point slots
    for that handful of anchors) -- not the whole call with 4 096-point slots for every environment.  Scores against the oracle,
    against the whole-pass retry (LCHD_NO_OVERFLOW_SUBSET), pass counts and store bytes."""
    import torch
    from loco_hd_amd.device import DeviceSession

    # (cluster_of_20000: the second pass takes the slots beyond 16 384 points -- unsorted collection + the global-memory row sort)
    n_sparse, n_cluster, n_cat = 60_000, (20_000 if variant == "cluster_of_20000" else 3_000), 7
    xa, ca, ta = _cluster_in_sparse_cloud(1, n_sparse, n_cluster, n_cat)
    xb, cb, tb = _cluster_in_sparse_cloud(2, n_sparse, n_cluster, n_cat)
    rng = np.random.default_rng(3)
    n_pairs = 6000
    pairs = np.stack([rng.integers(0, n_sparse, n_pairs), rng.integers(0, n_sparse, n_pairs)], 1).astype(np.int64)
    big = rng.choice(n_pairs, 90, replace=False)
    pairs[big[:30], 0] = n_sparse + rng.integers(0, n_cluster, 30)       # cluster anchor on side A
    pairs[big[30:60], 1] = n_sparse + rng.integers(0, n_cluster, 30)     # ... on side B
    pairs[big[60:], 0] = n_sparse + rng.integers(0, n_cluster, 30)       # ... on both
    pairs[big[60:], 1] = n_sparse + rng.integers(0, n_cluster, 30)
    cats = [f"c{i}" for i in range(n_cat)]
    kw, wf_idx = {}, None
    if variant == "same_object":
        xb, cb, tb = xa, ca, ta
    if variant == "tag_rule":
        kw["tag_pairing_rule"] = {"accept_same": False}  # (points carrying the anchor's tag -- 1 in 50 -- are left out)
    else:
        ta, tb = np.zeros_like(ta), np.zeros_like(tb)     # (the default rule keeps the points that carry the anchor's tag)

    def build(mod):
        wf = mod.WeightFunction("uniform", [3.0, 10.0])
        if variant == "weight_function_dictionary":
            wf = {"u": wf, "h": mod.WeightFunction("hyper_exp", [1.0, 0.2])}
        rule = mod.TagPairingRule(kw["tag_pairing_rule"]) if "tag_pairing_rule" in kw else None
        return mod.LoCoHD(cats, wf, rule)

    if variant == "weight_function_dictionary":
        wf_idx = (np.arange(n_pairs) % 2).astype(np.int32)
    lo = build(oracle)
    owfs = None
    if wf_idx is not None:
        owfs = lo._wfs(["u" if k == 0 else "h" for k in wf_idx], n_pairs)
    want = np.asarray(lo.from_arrays(xa, ca, ta, xb, cb, tb, pairs, 10.0, *(owfs if owfs else ())))

    def run(no_subset):
        if variant == "one_environment_per_workgroup":
            monkeypatch.setenv("LCHD_NO_ENV_GROUP", "1")  # (k_env_cells keeps the overflow list too)
        if no_subset:
            monkeypatch.setenv("LCHD_NO_OVERFLOW_SUBSET", "1")
        else:
            monkeypatch.delenv("LCHD_NO_OVERFLOW_SUBSET", raising=False)
        sess = DeviceSession(build(lh))
        a = sess.upload(xa, ca, ta)
        b = a if variant == "same_object" else sess.upload(xb, cb, tb)
        anchors = torch.from_numpy(pairs).cuda()
        wfi = None if wf_idx is None else torch.from_numpy(wf_idx).cuda()
        outs, counts = [], []
        for _ in range(2):
            before = sess.pass_counts()
            outs.append(sess.from_primitives(a, b, anchors, 10.0, wf_index=wfi).cpu().numpy())
            after = sess.pass_counts()
            counts.append({k: after[k] - before[k] if k != "store_bytes" else after[k] for k in after})
        sess.close()
        return outs, counts

    outs, counts = run(False)
    assert np.max(np.abs(outs[0] - want)) < 1e-11
    assert np.array_equal(outs[0], outs[1])
    for cnt in counts:  # every call: the first pass + ONE second pass over the cluster's pairs
        assert cnt["subset_passes"] == 1 and cnt["passes"] == 2
    whole, wcounts = run(True)
    assert np.max(np.abs(whole[0] - want)) < 1e-11
    assert np.max(np.abs(whole[0] - outs[0])) < 1e-11
    assert wcounts[0]["subset_passes"] == 0
    # the whole-pass retry gives EVERY environment of the call a 4 096-point slot; re-scoring the cluster's pairs adds the slots of
    # ~180 anchors to the first pass's store
    # (slots for min(atoms, pairs) anchors per side -- one side with twice as many when the sides are one object --, 8-byte keys + a
    #  category byte per point; a dictionary of two weight functions: the distances + one set of F keys per function)
    first_pass_store = 2 * n_pairs * 512 * (1 + 8 * (3 if variant == "weight_function_dictionary" else 1))
    assert counts[0]["store_bytes"] <= (2.5 if variant == "cluster_of_20000" else 1.3) * first_pass_store
    assert wcounts[0]["store_bytes"] >= 4 * counts[0]["store_bytes"]


def test_overflow_after_an_all_small_call_keeps_the_larger_pairs(lh, oracle):
    """A session whose previous call had only small pairs (sparse cloud: every pair fits the four-pairs-per-wavefront sweep) launches
    the next pass without the companion sweep for larger pairs.  When that next call has BOTH an overflowed cluster (second pass over
    its pairs) and medium environments (~170 points per side: more merged events than the team tile), the medium pairs must still be
    scored: the whole pass is repeated with the full launch set before the cluster's pairs are scored again (the reference has one
    code path for every environment size, src/locohd.rs:514-557)."""
    import torch
    from loco_hd_amd.device import DeviceSession

    n_cat = 6
    cats = [f"c{i}" for i in range(n_cat)]

    def cloud(seed):
        rng = np.random.default_rng(seed)
        n_sparse, n_medium, n_cluster = 40_000, 3_000, 2_500
        side = (n_sparse / 0.0025) ** (1 / 3)
        sparse = rng.uniform(0.0, side, (n_sparse, 3))
        medium = rng.uniform(0.0, 42.0, (n_medium, 3)) + np.array([-120.0, 0.0, 0.0])
        v = rng.normal(0.0, 1.0, (n_cluster, 3))
        cluster = np.array([0.0, -120.0, 0.0]) + v / np.linalg.norm(v, axis=1)[:, None] * (4.5 * rng.uniform(0, 1, (n_cluster, 1)) ** (1 / 3))
        xyz = np.concatenate([sparse, medium, cluster])
        return xyz, rng.integers(0, n_cat, len(xyz)).astype(np.int32), np.zeros(len(xyz), np.int32), n_sparse, n_medium, n_cluster

    xa, ca, ta, ns, nm, nc = cloud(11)
    xb, cb, tb, _, _, _ = cloud(12)
    rng = np.random.default_rng(13)
    n_pairs = 6000
    small = np.stack([rng.integers(0, ns, n_pairs), rng.integers(0, ns, n_pairs)], 1).astype(np.int64)
    mixed = small.copy()
    mixed[:800] = ns + rng.integers(0, nm, (800, 2))                     # medium environments on both sides
    mixed[800:860, 0] = ns + nm + rng.integers(0, nc, 60)                # cluster anchors (overflow their 512-point slots)
    mixed[860:900, 1] = ns + nm + rng.integers(0, nc, 40)
    rng.shuffle(mixed)
    lo = oracle.LoCoHD(cats, oracle.WeightFunction("hyper_exp", [1.0, 0.1]))
    want_small = np.asarray(lo.from_arrays(xa, ca, ta, xb, cb, tb, small, 10.0))
    want_mixed = np.asarray(lo.from_arrays(xa, ca, ta, xb, cb, tb, mixed, 10.0))
    sess = DeviceSession(lh.LoCoHD(cats, lh.WeightFunction("hyper_exp", [1.0, 0.1])))
    a, b = sess.upload(xa, ca, ta), sess.upload(xb, cb, tb)
    d_small, d_mixed = torch.from_numpy(small).cuda(), torch.from_numpy(mixed).cuda()
    for _ in range(2):  # the second call runs on the hint of the first: no companion sweep
        assert np.max(np.abs(sess.from_primitives(a, b, d_small, 10.0).cpu().numpy() - want_small)) < TIGHT
    out = torch.full((n_pairs,), -7.0, dtype=torch.float64, device="cuda")  # (a stale value would show)
    got = sess.from_primitives(a, b, d_mixed, 10.0, out=out).cpu().numpy()
    assert np.max(np.abs(got - want_mixed)) < TIGHT
    for _ in range(2):  # and the steady state of the mixed workload
        assert np.max(np.abs(sess.from_primitives(a, b, d_mixed, 10.0).cpu().numpy() - want_mixed)) < TIGHT
    assert np.max(np.abs(sess.from_primitives(a, b, d_small, 10.0).cpu().numpy() - want_small)) < TIGHT
    sess.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("LCHD_OVERFLOW_SEEDS", "6"))))  # (a one-off campaign: LCHD_OVERFLOW_SEEDS=200)
def test_overflowed_environments_random_configurations(lh, oracle, seed):
    """Random clouds with one to three dense clusters (600 ... 5000 points) in a sparse background, random thresholds, pair lists that
    touch the clusters on either side, category weights / Kolmogorov-Smirnov / a weight-function dictionary / a tag rule at random:
    whatever mixture of first pass, second pass over the overflowed environments' pairs and whole-pass retry the call takes, the scores
    are the oracle's."""
    import torch
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(900 + seed)
    n_cat = int(rng.choice([4, 9, 14, 23]))
    cats = [f"c{i}" for i in range(n_cat)]

    def cloud():
        n_sparse = int(rng.integers(8000, 20000))
        side = (n_sparse / 0.003) ** (1 / 3)
        parts = [rng.uniform(0.0, side, (n_sparse, 3))]
        for _ in range(int(rng.integers(1, 4))):
            m = int(rng.integers(600, 5000))
            v = rng.normal(0.0, 1.0, (m, 3))
            parts.append(rng.uniform(0.2 * side, 0.8 * side, 3) + v / np.linalg.norm(v, axis=1)[:, None] * (rng.uniform(3.0, 7.0) * rng.uniform(0, 1, (m, 1)) ** (1 / 3)))
        xyz = np.concatenate(parts)
        return xyz, rng.integers(0, n_cat, len(xyz)).astype(np.int32), rng.integers(0, 30, len(xyz)).astype(np.int32), n_sparse

    xa, ca, ta, sa_ = cloud()
    xb, cb, tb, sb_ = cloud()
    n_pairs = 3000
    pairs = np.stack([rng.integers(0, len(xa), n_pairs), rng.integers(0, len(xb), n_pairs)], 1).astype(np.int64)
    k = int(rng.integers(5, 200))
    pairs[:k, 0] = rng.integers(sa_, len(xa), k)       # cluster anchors on side A
    pairs[k:2 * k, 1] = rng.integers(sb_, len(xb), k)  # ... on side B
    rng.shuffle(pairs)
    thr = float(rng.uniform(6.0, 12.0))
    kind = int(rng.integers(0, 5))
    kw = {}
    if kind == 1:
        kw["category_weights"] = rng.uniform(0.3, 2.5, n_cat).tolist()
    use_rule = kind == 3
    if not use_rule:
        ta, tb = np.zeros_like(ta), np.zeros_like(tb)

    def build(mod):
        wf = mod.WeightFunction("hyper_exp", [1.0, 0.12])
        if kind == 4:
            wf = {"a": wf, "b": mod.WeightFunction("uniform", [2.0, 9.0]), "c": mod.WeightFunction("kumaraswamy", [1.0, 11.0, 1.5, 2.5])}
        if kind == 2:
            kw["statistical_distance"] = mod.StatisticalDistance("Kolmogorov-Smirnov", [])
        return mod.LoCoHD(cats, wf, mod.TagPairingRule({"accept_same": False}) if use_rule else None, **kw)

    wf_idx = rng.integers(0, 3, n_pairs).astype(np.int32) if kind == 4 else None
    lo = build(oracle)
    extra = lo._wfs(["abc"[i] for i in wf_idx], n_pairs) if kind == 4 else ()
    want = np.asarray(lo.from_arrays(xa, ca, ta, xb, cb, tb, pairs, thr, *extra))
    sess = DeviceSession(build(lh))
    a, b = sess.upload(xa, ca, ta), sess.upload(xb, cb, tb)
    anchors = torch.from_numpy(pairs).cuda()
    wfi = None if wf_idx is None else torch.from_numpy(wf_idx).cuda()
    for _ in range(2):
        got = sess.from_primitives(a, b, anchors, thr, wf_index=wfi).cpu().numpy()
        assert np.max(np.abs(got - want)) < 1e-11, (seed, kind, thr)
    sess.close()
