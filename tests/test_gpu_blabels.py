"""Row f4 of SURVEY.md section 8: scores -> B labels -> PrimitiveAssigner.generate_primitive_pdb
(/root/reference/loco_hd/atom_converter_utils.py:133-168; callers python_codes/ensembles/compare_ensembles.py:196-323,
python_codes/trajectory_analyzer.py:97-120), through the `loco_hd` import name, against the committed oracle texts
tests/golden/blabelled_{ensemble,trajectory}.pdb (made by tests/golden/make_blabelled_pdb.py) BYTE FOR BYTE."""
import io
from pathlib import Path

import numpy as np
import pytest

import blabel_case as bc
import pdb_util

GOLD = Path(__file__).resolve().parent / "golden"


def _product_templates(tmp_path):
    import loco_hd  # the reference's import name (loco_hd/__init__.py:1-2)
    from loco_hd_amd.pdb_reader import PDBParser  # stands in for Bio.PDB.PDBParser (absent here; compare_ensembles.py:203)

    pa = loco_hd.PrimitiveAssigner(pdb_util.write_scheme(tmp_path / "scheme.json"))
    models = [PDBParser(QUIET=True).get_structure("", io.StringIO(t))[0] for t in bc.conformer_texts()]
    return loco_hd, pa, models, [pa.assign_primitive_structure(m) for m in models]


def test_golden_texts_are_the_oracle_chain(oracle, tmp_path):
    """CPU: the committed texts are what the oracle chain produces today (the maker script and the fixtures stay in step), and the
    product's writer -- pure formatting -- gives the same bytes when it is handed the oracle's labels."""
    from oracle import atom_converter_oracle as aco

    from loco_hd_amd import PrimitiveAssigner
    from loco_hd_amd.pdb_reader import PDBParser

    scheme_path = pdb_util.write_scheme(tmp_path / "scheme.json")
    scheme = aco.load_scheme(scheme_path)
    types = list(pdb_util.TEST_SCHEME)
    texts = bc.conformer_texts()
    templates = [aco.assign_primitive_structure(scheme, aco.read_pdb(t, "s")[0]) for t in texts]
    b_ens = bc.ensemble_b_labels(oracle, types, [[(pt, c, fid) for pt, c, (fid, _rn, _nm) in tl] for tl in templates])
    want = (GOLD / "blabelled_ensemble.pdb").read_text()
    assert aco.generate_primitive_pdb(types, templates[0], b_ens) == want
    pa = PrimitiveAssigner(scheme_path)
    assert pa.all_primitive_types == types
    prats = pa.assign_primitive_structure(PDBParser(QUIET=True).get_structure("", io.StringIO(texts[0]))[0])
    assert pa.generate_primitive_pdb(prats, b_ens) == want
    assert pa.generate_primitive_pdb(prats, list(b_ens)) == want


@pytest.mark.gpu
def test_ensemble_scores_as_b_labels_byte_for_byte(tmp_path):
    """compare_ensembles.py:196-323 with the HIP from_dmxs: per-atom mean over all conformer pairs into the B column."""
    loco_hd, pa, _models, tls = _product_templates(tmp_path)
    triples = [[(p.primitive_type, p.coordinates, p.atom_source.source_residue) for p in tl] for tl in tls]
    b = bc.ensemble_b_labels(loco_hd, pa.all_primitive_types, triples)
    got = pa.generate_primitive_pdb(tls[0], b_labels=b)
    assert got == (GOLD / "blabelled_ensemble.pdb").read_text()


@pytest.mark.gpu
def test_trajectory_scores_as_b_labels_byte_for_byte(tmp_path):
    """BASELINE config 4's pipeline at a small size: float32 source-atom frames -> device centroids -> per-frame scores of the
    "Cent" anchors against the reference frame (streamed frames buffer) -> mean over the frames into the B column."""
    loco_hd, pa, models, tls = _product_templates(tmp_path)
    from loco_hd_amd.device import DeviceSession

    topo = pa.compile_topology(models[0])
    frames = np.stack([pa.compile_topology(m).atom_coords for m in models[1:]]).astype(np.float32)
    cent = [i for i, t in enumerate(topo.primitive_types) if t == "Cent"]
    lchd = loco_hd.LoCoHD(pa.all_primitive_types, loco_hd.WeightFunction(*bc.WF), loco_hd.TagPairingRule({"accept_same": False}))
    interner = {}
    packed = topo.pack(lchd, interner)
    sess = DeviceSession(lchd, interner=interner)
    ref = sess.upload(packed.xyz, packed.cat, packed.tag)
    scores = sess.score_trajectory(ref, frames, np.stack([cent, cent], 1), 10.0, chunk=3, topology=topo)
    sess.close()
    b = np.zeros(len(tls[0]))
    b[cent] = np.mean(scores, axis=0)
    got = pa.generate_primitive_pdb(tls[0], b_labels=b)
    assert got == (GOLD / "blabelled_trajectory.pdb").read_text()
