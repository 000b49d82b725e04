"""Host logic of the product package on a CPU-only machine: validation, error classes, string interning, the
host-side leaves of the C ABI (CDFs, statistical distances).  Mirrors the reference's own unit tests
(/root/reference/tests/test_wfs.py, test_tag_pairing_rule.py:8-98, test_locohd.py:54-73) against loco_hd_amd."""
import itertools

import numpy as np
import pytest

import loco_hd_amd as lh
from test_oracle_kat import WF_ERRS, WF_KATS


@pytest.mark.parametrize("name,params,cases", WF_KATS)
def test_weight_function_kats(name, params, cases):  # tests/test_wfs.py:8-138
    wf = lh.WeightFunction(name, params)
    for a, b, want in cases:
        assert wf.integral_range(float(a), float(b)) == pytest.approx(want, abs=5e-5)
    assert wf.function_name == name and wf.parameters == [float(p) for p in params]


@pytest.mark.parametrize("name,params", WF_ERRS)
def test_weight_function_errors(name, params):  # tests/test_wfs.py:29-156
    with pytest.raises(ValueError):
        lh.WeightFunction(name, params)


def test_weight_function_misc(oracle):
    with pytest.raises(ValueError):
        lh.WeightFunction("gauss", [1.0])
    wf = lh.WeightFunction("dagum", [1.7, 2.5, 9.0])
    with pytest.raises(ValueError):  # weight_function.rs:97-100
        wf.integral_point(-1e-9)
    xs = np.linspace(0.0, 40.0, 201)
    ref = oracle.WeightFunction("dagum", [1.7, 2.5, 9.0])
    assert np.allclose(wf.integral_vec(xs), ref.integral_vec(xs), rtol=0, atol=1e-15)
    assert wf.integral_point(float("inf")) == 1.0 and wf.integral_point(0.0) == 0.0


def test_statistical_distance_leaves(oracle):
    rng = np.random.default_rng(3)
    for name, prm in (("Hellinger", [2.0]), ("Hellinger", [3.3]), ("Kolmogorov-Smirnov", []), ("Kullback-Leibler", [0.5]),
                      ("Renyi", [2.4, 0.7]), ("Renyi", [1.0, 0.5]), ("Renyi", [0.0, 0.1]), ("Renyi", [float("inf"), 0.2])):
        a, b = lh.StatisticalDistance(name, prm), oracle.StatisticalDistance(name, prm)
        for _ in range(20):
            p, q = rng.dirichlet(np.ones(7)), rng.dirichlet(np.ones(7))
            assert a.run(p, q) == pytest.approx(b.run(p, q), abs=1e-15)
    for bad in (("Hellinger", []), ("Hellinger", [1.0, 2.0]), ("Kolmogorov-Smirnov", [1.0]), ("Renyi", [1.0]), ("Wasserstein", [])):
        with pytest.raises(ValueError):
            lh.StatisticalDistance(*bad)


def test_tag_pairing_truth_table():  # tests/test_tag_pairing_rule.py:8-98
    tpr = lh.TagPairingRule({"accept_same": True})
    assert tpr.pair_accepted(("A", "A")) and not tpr.pair_accepted(("A", "B"))
    tpr = lh.TagPairingRule({"accept_same": False})
    assert not tpr.pair_accepted(("A", "A")) and tpr.pair_accepted(("A", "B"))
    listed = {("A", "B"), ("A", "C"), ("B", "C")}
    for accepted_pairs, ordered in itertools.product([True, False], [True, False]):
        tpr = lh.TagPairingRule({"tag_pairs": listed, "accepted_pairs": accepted_pairs, "ordered": ordered})
        for pair in itertools.product("ABC", repeat=2):
            hit = pair in listed or (not ordered and pair[::-1] in listed)
            assert tpr.pair_accepted(pair) == (hit if accepted_pairs else not hit)
        assert "WithList" in tpr.get_dbg_str()
    with pytest.raises(TypeError):
        lh.TagPairingRule({"something": 1})


def test_locohd_constructor():  # tests/test_locohd.py:54-73, src/locohd.rs:289-389
    wf = lh.WeightFunction("uniform", [0.0, 4.0])
    types = ["O", "A", "B", "C"]
    with pytest.raises(ValueError):
        lh.LoCoHD([], wf)
    for bad in ([1.0, 1.0, 1.0], [1.0] * 5, [1.0, -1.0, 1.0, 1.0], [1.0, 0.0, 1.0, 1.0]):
        with pytest.raises(ValueError):
            lh.LoCoHD(types, wf, category_weights=bad)
    lchd = lh.LoCoHD(types, wf, lh.TagPairingRule({"accept_same": False}), 4)  # positional use, README.md:373-378
    assert lchd.categories == {"O": 0, "A": 1, "B": 2, "C": 3}
    assert lchd.category_weights == [1.0] * 4 and lchd.w_func is wf
    dflt = lh.LoCoHD(types)  # defaults :349-370
    assert dflt.w_func.function_name == "uniform" and dflt.w_func.parameters == [3.0, 10.0]
    assert dflt.tag_pairing_rule.pair_accepted(("x", "x"))
    assert lh.LoCoHD(["A", "B", "A"]).categories == {"A": 2, "B": 1}  # HashMap collect keeps the last index (:312-316)


def test_weight_function_key_rules():  # src/locohd.rs:230-283
    single = lh.LoCoHD(["A"], lh.WeightFunction("uniform", [0.0, 4.0]))
    multi = lh.LoCoHD(["A"], {"x": lh.WeightFunction("uniform", [0.0, 4.0]), "y": lh.WeightFunction("uniform", [1.0, 2.0])})
    assert single._wf_indices(None, 5) is None
    assert multi._wf_indices(["y", "x", "y"], 3).tolist() == [1, 0, 1]
    for lchd, keys, n in ((single, ["x"], 1), (multi, None, 1), (multi, ["x"], 2), (multi, ["x", "nope"], 2)):
        with pytest.raises(ValueError):
            lchd._wf_indices(keys, n)


def test_anchor_pair_parsing():  # AnchorPairSpecifier, src/locohd.rs:34-40
    split = lh.LoCoHD._split_anchor_pairs
    assert split([(0, 1), (2, 3)]) == ([(0, 1), (2, 3)], None)
    assert split([(0, 1, "k")]) == ([(0, 1)], ["k"])
    assert split([]) == ([], [])  # the empty list is the with-key variant
    with pytest.raises(TypeError):
        split([(0, 1), (1, 2, "k")])
    with pytest.raises(OverflowError):
        split([(0, -1)])


def test_packing_interns_like_the_reference():
    lchd = lh.LoCoHD(["A", "B"])
    prims = [lh.PrimitiveAtom("A", "r1", [0, 0, 0]), lh.PrimitiveAtom("Z", "r2", (1, 2, 3)), lh.PrimitiveAtom("B", "r1", np.ones(3))]
    interner = {}
    p = lchd.pack(prims, interner)
    assert p.cat.tolist() == [0, -1, 1] and p.tag.tolist() == [0, 1, 0] and p.xyz[1].tolist() == [1.0, 2.0, 3.0]
    prims[0].coordinates = [9, 9, 9]
    assert prims[0].coordinates == [9.0, 9.0, 9.0]
    with pytest.raises(ValueError):
        lh.PrimitiveAtom("A", "", [1.0, 2.0])


def test_scoring_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lchd = lh.LoCoHD(["A"])
    with pytest.raises(lh.DeviceError):
        lchd.from_anchors(["A"], ["A"], [0.0], [0.0])
    with pytest.raises(lh.DeviceError):
        lchd.from_coords(["A"], ["A"], [[0.0, 0.0, 0.0]], [[0.0, 0.0, 0.0]])


def test_native_pack_equals_python_pack():
    """loco_hd_amd/_fastpack (CPython helper, the counterpart of the reference's PyO3 argument extraction) fills the same
    arrays and interns the same tags as the pure-Python conversion loop."""
    import numpy as np

    import loco_hd_amd as lh
    from loco_hd_amd import api

    assert api._fastpack is not None, "the _fastpack extension was not built (make -C loco_hd_amd/csrc)"
    rng = np.random.default_rng(3)
    cats = ["a", "b", "c"]
    prims = [lh.PrimitiveAtom(cats[i % 3] if i % 7 else "zz", f"T{i // 4}", rng.uniform(-5, 5, 3)) for i in range(200)]
    prims[3].primitive_type = np.str_("b")
    prims[4].coordinates = np.asarray([1, 2, 3], dtype=np.float32)

    class Duck:  # any object with the three attributes is accepted, like in the Python loop
        def __init__(self, t, g, c):
            self.primitive_type, self.tag, self.coordinates = t, g, c

    prims.append(Duck("c", "T0", (7, 8, 9)))
    lchd = lh.LoCoHD(cats)
    i_native, i_python = {"seed": 0}, {"seed": 0}
    native = lchd.pack(prims, i_native)
    saved, api._fastpack = api._fastpack, None
    try:
        python = lchd.pack(prims, i_python)
    finally:
        api._fastpack = saved
    assert np.array_equal(native.xyz, python.xyz) and np.array_equal(native.cat, python.cat) and np.array_equal(native.tag, python.tag)
    assert i_native == i_python and native.cat.dtype == np.int32 and native.xyz.shape == (201, 3)
    assert (native.cat == -1).sum() == sum(1 for i in range(200) if i % 7 == 0)
    import pytest

    with pytest.raises(ValueError):
        lchd.pack([Duck("a", "t", (1, 2))], {})
    with pytest.raises(AttributeError):
        lchd.pack([object()], {})


def test_primitive_atom_record_and_native_identity_helpers():
    """primitive_atom.rs:4-25: get + set attributes; the setters also stamp the module's mutation counter (LoCoHD's packed-list
    cache), construction does not.  _fastpack.items_tuple / same_items compare by identity."""
    import pickle

    import loco_hd_amd as lh
    from loco_hd_amd import api

    before = api._ATOM_MUTATIONS[0]
    a = lh.PrimitiveAtom("A", "t", (1, 2, 3))
    assert (a.primitive_type, a.tag, a.coordinates) == ("A", "t", [1.0, 2.0, 3.0]) and api._ATOM_MUTATIONS[0] == before
    a.primitive_type, a.tag, a.coordinates = "B", "u", [4, 5, 6]
    assert (a.primitive_type, a.tag, a.coordinates) == ("B", "u", [4.0, 5.0, 6.0]) and api._ATOM_MUTATIONS[0] == before + 3
    c = a.coordinates
    c[0] = 99.0  # a copy, as in the reference (the getter clones)
    assert a.coordinates[0] == 4.0
    with pytest.raises(ValueError):
        a.coordinates = [1.0, 2.0]
    with pytest.raises(ValueError):
        lh.PrimitiveAtom("A", "t", [1.0])
    b = pickle.loads(pickle.dumps(a))
    assert (b.primitive_type, b.tag, b.coordinates) == ("B", "u", [4.0, 5.0, 6.0])
    if api._fastpack is not None:
        items = [lh.PrimitiveAtom("A", "", [0, 0, i]) for i in range(5)]
        tup = api._fastpack.items_tuple(items, lh.PrimitiveAtom)
        assert isinstance(tup, tuple) and all(x is y for x, y in zip(tup, items))
        assert api._fastpack.same_items(items, tup) and api._fastpack.same_items(tuple(items), tup)
        assert not api._fastpack.same_items(items[:4], tup) and not api._fastpack.same_items(items[::-1], tup)
        assert api._fastpack.items_tuple(items + ["x"], lh.PrimitiveAtom) is None
        assert api._fastpack.items_tuple(np.asarray(items, dtype=object), lh.PrimitiveAtom) is None  # lists / tuples only
        assert api._fastpack.items_tuple([(1, 2), (3, 4)], tuple) is not None and api._fastpack.items_tuple([(1, 2), [3, 4]], tuple) is None


def test_interned_ids_pack_like_the_general_path():
    """_fastpack.pack_atoms (a gather over the ids every PrimitiveAtom interns when it is constructed or changed) against pack_into (the
    strings looked up per atom): same arrays for plain lists and tuples, after setters, after pickling, with labels that are not str;
    lists that hold anything but exactly PrimitiveAtom fall back to the general path (primitive_atom.rs:4-25: get + set attributes)."""
    import pickle

    from loco_hd_amd import api

    cats = ["A", "B", "C", "7"]
    lchd = lh.LoCoHD(cats)
    rng = np.random.default_rng(3)
    atoms = [lh.PrimitiveAtom(cats[i % 4], f"r{i // 3}", rng.uniform(0, 9, 3)) for i in range(50)]
    atoms[5] = lh.PrimitiveAtom(np.str_("B"), np.str_("r1"), (1, 2, 3))      # NumPy strings, integer coordinates
    atoms[6] = lh.PrimitiveAtom(7, "r2", [0.0, 0.0, 0.0])                      # a label that is not a str: str(label) is looked up
    atoms[7] = lh.PrimitiveAtom("nowhere", "r2", [0.0, 0.0, 0.0])              # not in the map: -1 (raises when the environment is used)

    def both(seq):
        fast, general = lchd._pack_global(seq), lchd.pack(seq, api._TAG_IDS)
        assert np.array_equal(fast.xyz, general.xyz) and np.array_equal(fast.cat, general.cat) and np.array_equal(fast.tag, general.tag)
        return fast

    first = both(atoms)
    assert first.cat[5] == 1 and first.cat[6] == 3 and first.cat[7] == -1 and first.tag[5] == first.tag[4]
    both(tuple(atoms))
    atoms[0].tag, atoms[1].primitive_type, atoms[2].coordinates = "elsewhere", "C", [9.0, 8.0, 7.0]
    second = both(atoms)
    assert second.tag[0] != first.tag[0] and second.cat[1] == 2 and second.xyz[2].tolist() == [9.0, 8.0, 7.0]
    both(pickle.loads(pickle.dumps(atoms)))
    other = lh.LoCoHD(["C", "B"])   # another instance, another category map over the same type ids
    assert other._pack_global(atoms).cat[:8].tolist() == [-1 if c not in ("B", "C") else ["C", "B"].index(c) for c in
                                                          [str(a.primitive_type) for a in atoms[:8]]]

    class Sub(lh.PrimitiveAtom):
        pass

    mixed = atoms[:10] + [Sub("A", "r0", [1.0, 1.0, 1.0])]
    xyz, cat, tag = np.empty((11, 3)), np.empty(11, np.int32), np.empty(11, np.int32)
    assert api._fastpack.pack_atoms(mixed, lh.PrimitiveAtom, lchd._type_map(), xyz, cat, tag) is False
    both(mixed)  # ... and the fall-back gives the same ids


def test_native_anchor_pair_extraction():
    """_fastpack.pairs_into = AnchorPairSpecifier (src/locohd.rs:34-40): 3-tuples first (an empty list is that variant), then 2-tuples;
    mixed lengths are a TypeError, negative indices an OverflowError (usize)."""
    from loco_hd_amd import api

    lchd = lh.LoCoHD(["A"], {"u": lh.WeightFunction("uniform", [1.0, 2.0]), "v": lh.WeightFunction("uniform", [1.0, 3.0])})
    arr, idx = lchd._anchor_arrays([(1, 2, "v"), (np.int64(3), 4, "u")])
    assert arr.tolist() == [[1, 2], [3, 4]] and idx.tolist() == [1, 0]
    single = lh.LoCoHD(["A"])
    arr, idx = single._anchor_arrays(((5, 6), [7, 8]))
    assert arr.tolist() == [[5, 6], [7, 8]] and idx is None
    with pytest.raises(ValueError):
        single._anchor_arrays([])  # the with-key variant against a single weight function (src/locohd.rs:276-281)
    with pytest.raises(TypeError):
        single._anchor_arrays([(1, 2), (1, 2, "u")])
    with pytest.raises(TypeError):
        single._anchor_arrays([(1, 2, 3, 4)])
    with pytest.raises(OverflowError):
        single._anchor_arrays([(1, -2)])
    with pytest.raises(TypeError):
        single._anchor_arrays([(1.5, 2)])  # usize extraction does not take floats
