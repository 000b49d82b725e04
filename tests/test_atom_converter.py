"""Structure -> primitive-atom step, CPU side: product code vs the oracle restatement.

The reference has no test for loco_hd/atom_converter_utils.py or loco_hd/__main__.py, so these are modelled on its use in
README.md:250-330 and python_codes/trajectory_analyzer.py:37-74.
"""
import io
from pathlib import Path

import numpy as np
import pytest

from loco_hd_amd import PrimitiveAssigner, PrimitiveAtomTemplate, prat_to_pra
from loco_hd_amd.__main__ import parse_anchor_pairing, parse_cli_args, pra_template_list_to_idx_dict
from loco_hd_amd.pdb_reader import PDBParser
from oracle import atom_converter_oracle as aco
from tests import pdb_util

REF_SCHEMES = sorted(Path("/root/reference/primitive_typings").glob("*.config.json"))


def _as_tuples(model):
    return [(r.full_id, r.resname, [(a.name, a.coord) for a in r.get_atoms()]) for r in model.get_residues()]


def _same_residues(got, want):
    assert [(g[0], g[1], [n for n, _ in g[2]]) for g in got] == [(w[0], w[1], [n for n, _ in w[2]]) for w in want]
    for g, w in zip(got, want):
        for (_, cg), (_, cw) in zip(g[2], w[2]):
            assert cg.dtype == np.float32 and cg.tobytes() == cw.tobytes()


@pytest.mark.parametrize("seed", range(6))
def test_reader_matches_independent_restatement(seed):
    text = pdb_util.synthetic_pdb(seed=seed, n_res=30, models=1 + seed % 3)
    st = PDBParser(QUIET=True).get_structure("s", io.StringIO(text))
    want = aco.read_pdb(text, "s")
    assert [m.id for m in st] == sorted(want)
    for m in st:
        _same_residues(_as_tuples(m), want[m.id])


def test_reader_corner_cases():
    text = pdb_util.nasty_pdb()
    st = PDBParser(QUIET=True).get_structure("x", io.StringIO(text))
    _same_residues(_as_tuples(st[0]), aco.read_pdb(text, "x")[0])
    a, b = st[0]["A"], st[0]["B"]
    r1 = a[1]
    assert r1.resname == "ALA"
    # duplicate CA dropped; CB picks altloc B (occupancy 0.7); C keeps A on a tie; O (blank first, then altloc B) is
    # re-added behind C with the occupancy-1.0 blank copy selected; "CA  " stays apart from " CA "
    assert [x.name for x in r1.get_atoms()] == ["N", "CA", "CB", "C", "O", "CA  "]
    assert r1["CA"].coord.tolist() == [1, 0, 0] and r1["CB"].coord.tolist() == [2, 1, 0] and r1["C"].coord.tolist() == [3, 0, 0]
    assert r1["O"].altloc == " " and r1["O"].coord.tolist() == [4, 0, 0] and r1["CA  "].coord.tolist() == [5, 0, 0]
    assert a[3].resname == "THR" and [x.name for x in a[3].get_atoms()] == ["N", "OG1"]          # last variant selected
    assert [x.name for x in a[4].get_atoms()] == ["N", "CA"]                                        # continued after chain B
    assert a[(" ", 4, "A")].full_id == ("x", 0, "A", (" ", 4, "A"))
    assert a[("W", 5, " ")].resname == "HOH" and [x.name for x in a[("H_LIG", 6, " ")].get_atoms()] == ["C1"]
    assert [x.name for x in a[8].get_atoms()] == ["N B "]
    assert [r.resname for r in b] == ["LYS"]
    assert [(r.id[0], r.id[1], r.id[2]) for r in a] == [(" ", 1, " "), (" ", 3, " "), (" ", 4, " "), (" ", 4, "A"), ("W", 5, " "),
                                                        ("H_LIG", 6, " "), (" ", 7, " "), (" ", 8, " ")]


def test_model_selection_and_missing_model():
    text = pdb_util.synthetic_pdb(seed=3, n_res=5, models=3, hetero=False)
    st = PDBParser().get_structure("s", io.StringIO(text))
    assert len(st) == 3 and st[2].id == 2
    with pytest.raises(KeyError):
        st[3]


def _check_assign(scheme_path, text):
    st = PDBParser(QUIET=True).get_structure("s1", io.StringIO(text))[0]
    pa = PrimitiveAssigner(scheme_path)
    got = pa.assign_primitive_structure(st)
    want = aco.assign_primitive_structure(aco.load_scheme(scheme_path), aco.read_pdb(text, "s1")[0])
    assert len(got) == len(want) and len(got) > 0
    for g, (ptype, cen, (full_id, resname, names)) in zip(got, want):
        assert isinstance(g, PrimitiveAtomTemplate)
        assert (g.primitive_type, g.atom_source.source_residue, g.atom_source.source_residue_name, g.atom_source.source_atom) == \
               (ptype, full_id, resname, names)
        assert np.asarray(g.coordinates).tobytes() == np.asarray(cen).tobytes()  # same np.mean call, bit for bit
    # the compiled topology gives the same primitive atoms and the same float32 centroids
    topo = pa.compile_topology(st) if all(len(w[2][2]) for w in want) else None
    if topo is not None:
        assert topo.primitive_types == [w[0] for w in want]
        assert topo.centroids().tobytes() == np.stack([np.asarray(w[1], dtype=np.float32) for w in want]).tobytes()
        assert topo.tags == [prat_to_pra(g).tag for g in got]
    return pa, got, want


@pytest.mark.parametrize("seed", range(4))
def test_assign_matches_oracle_on_test_scheme(tmp_path, seed):
    scheme = pdb_util.write_scheme(tmp_path / "scheme.json")
    pa, got, _ = _check_assign(scheme, pdb_util.synthetic_pdb(seed=seed, n_res=50))
    assert pa.all_primitive_types == list(pdb_util.TEST_SCHEME)
    assert "Never" not in {g.primitive_type for g in got} and "Wat" in {g.primitive_type for g in got}


@pytest.mark.skipif(not REF_SCHEMES, reason="the reference's typing schemes are only present in the build container")
@pytest.mark.parametrize("scheme", REF_SCHEMES, ids=lambda p: p.name)
def test_assign_matches_oracle_on_reference_schemes(scheme):
    """The four typing schemes the reference ships (primitive_typings/*.config.json), read in place as data."""
    pa, got, _ = _check_assign(scheme, pdb_util.synthetic_pdb(seed=11, n_res=80, hetero=False))
    assert set(pa.all_primitive_types) >= {g.primitive_type for g in got}


def test_topology_centroids_for_many_frames(tmp_path):
    scheme = pdb_util.write_scheme(tmp_path / "scheme.json")
    text = pdb_util.synthetic_pdb(seed=5, n_res=40)
    st = PDBParser().get_structure("s", io.StringIO(text))[0]
    pa = PrimitiveAssigner(scheme)
    topo = pa.compile_topology(st)
    rng = np.random.default_rng(0)
    frames = (topo.atom_coords[None] + rng.normal(0, 0.5, (7,) + topo.atom_coords.shape)).astype(np.float32)
    single = next(p for p in range(len(topo)) if topo.src_start[p + 1] - topo.src_start[p] == 1)
    frames[3, topo.src_idx[topo.src_start[single]]] = -0.0  # np.mean([-0.0]) is +0.0: add.reduce starts from +0
    cen = topo.centroids(frames)
    assert cen.shape == (7, len(topo), 3) and cen.dtype == np.float32
    for f in range(7):
        for p in range(len(topo)):
            members = [frames[f, i] for i in topo.src_idx[topo.src_start[p]:topo.src_start[p + 1]]]
            assert np.mean(members, axis=0).tobytes() == cen[f, p].tobytes()


def test_empty_any_group_is_refused_by_the_topology(tmp_path):
    scheme = pdb_util.write_scheme(tmp_path / "s.json", {"X": [[".+", "ZZ", "any"]]})
    st = PDBParser().get_structure("s", io.StringIO(pdb_util.synthetic_pdb(seed=1, n_res=3, hetero=False)))[0]
    pa = PrimitiveAssigner(scheme)
    with pytest.warns(RuntimeWarning):  # the reference's np.mean([]) warns and yields NaN (atom_converter_utils.py:126)
        prats = pa.assign_primitive_structure(st)
    assert len(prats) == 6 and all(np.isnan(p.coordinates).all() for p in prats)
    with pytest.raises(ValueError):
        pa.compile_topology(st)


def test_generate_primitive_pdb_matches_oracle(tmp_path):
    scheme = pdb_util.write_scheme(tmp_path / "scheme.json")
    text = pdb_util.synthetic_pdb(seed=2, n_res=25)
    st = PDBParser().get_structure("s", io.StringIO(text))[0]
    pa = PrimitiveAssigner(scheme)
    prats = pa.assign_primitive_structure(st)
    want_t = aco.assign_primitive_structure(aco.load_scheme(scheme), aco.read_pdb(text, "s")[0])
    b = np.random.default_rng(1).uniform(0, 99, len(prats))
    for labels in (None, b, list(b)):
        got = pa.generate_primitive_pdb(prats, labels)
        assert got == aco.generate_primitive_pdb(pa.all_primitive_types, want_t, labels)
    line = pa.generate_primitive_pdb(prats).splitlines()[0]
    assert len(line) == 80 and line.startswith("ATOM      1    A ") and line.endswith("Pr  ")
    with pytest.raises(Exception):
        pa.all_primitive_types = ["x"]


def test_anchor_pairing_file_format():
    txt = " /2-GLU/OE1,OE2:B/73-CYS/SG;\n /6-ARG/CZ:B/82-ILE/CB,CG1,CG2,CD1"
    pairs = parse_anchor_pairing(txt.replace("\n", "").split(";"))
    assert pairs[0] == ((" ", "2-GLU", frozenset({"OE1", "OE2"})), ("B", "73-CYS", frozenset({"SG"})))
    assert pairs[1][1] == ("B", "82-ILE", frozenset({"CB", "CG1", "CG2", "CD1"}))
    with pytest.raises(ValueError):
        parse_anchor_pairing(["A/1-GLY/CA:B/2-GLY/CA", ""])  # trailing semicolon -> empty entry, as in the reference


def test_template_index_and_cli_defaults(tmp_path):
    scheme = pdb_util.write_scheme(tmp_path / "scheme.json")
    st = PDBParser().get_structure("s", io.StringIO(pdb_util.synthetic_pdb(seed=4, n_res=10, hetero=False, altlocs=False)))[0]
    prats = PrimitiveAssigner(scheme).assign_primitive_structure(st)
    idx = pra_template_list_to_idx_dict(prats)
    k = next(i for i, p in enumerate(prats) if p.primitive_type == "Pos" or p.primitive_type == "Bb")
    src = prats[k].atom_source
    assert idx[(src.source_residue[2], f"{src.source_residue[3][1]}-{src.source_residue_name}", frozenset(src.source_atom))] == k
    pra = prat_to_pra(prats[k])
    assert pra.tag == f"{src.source_residue[2]}/{src.source_residue[3][1]}-{src.source_residue_name}"
    assert pra.coordinates == [float(x) for x in prats[k].coordinates]
    args = parse_cli_args(["-s1", "a.pdb", "-s2", "b.pdb", "-pts", "t.json", "-apf", "p.txt"])
    assert args.model_number == 0 and args.upper_distance_cutoff == 10.0 and args.number_of_threads is None
    assert args.tag_pairing_rule_args == {"accept_same": False}
    assert args.weight_function_args == {"function_name": "uniform", "parameters": [3.0, 10.0]}
