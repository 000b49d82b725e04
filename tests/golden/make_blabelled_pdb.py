"""Writes tests/golden/blabelled_{ensemble,trajectory}.pdb: the B-factor-labelled primitive-atom PDB texts of the two cases of
tests/blabel_case.py, produced by the CPU ORACLE chain alone (oracle/atom_converter_oracle.py: reader, assigner and the piecewise
restatement of generate_primitive_pdb, /root/reference/loco_hd/atom_converter_utils.py:133-168; oracle/locohd_oracle.c for the scores).

    python tests/golden/make_blabelled_pdb.py          (from the repository root; no GPU, no reference checkout needed)

Provenance: the reference holds no fixture for this writer and its Rust core cannot be built here, so the
expected texts are oracle-generated, like tests/golden/oracle_outputs.json.  tests/test_gpu_blabels.py runs the same two recipes
through the product (`import loco_hd`, HIP kernels) and compares byte for byte.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

import blabel_case as bc  # noqa: E402
import pdb_util  # noqa: E402
from oracle import atom_converter_oracle as aco  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    orc.build()
    out = Path(__file__).resolve().parent
    scheme_path = pdb_util.write_scheme(out / "_scheme.tmp.json")
    scheme = aco.load_scheme(scheme_path)
    scheme_path.unlink()
    types = list(pdb_util.TEST_SCHEME)  # the order of the typing scheme (all_primitive_types)
    templates = [aco.assign_primitive_structure(scheme, aco.read_pdb(t, "s")[0]) for t in bc.conformer_texts()]
    triples = [[(pt, c, fid) for pt, c, (fid, _rn, _nm) in tl] for tl in templates]
    # ensemble: mean from_dmxs score per primitive atom over all conformer pairs
    b_ens = bc.ensemble_b_labels(orc, types, triples)
    (out / "blabelled_ensemble.pdb").write_text(aco.generate_primitive_pdb(types, templates[0], b_ens))
    # trajectory: conformer 0 against the others, "Cent" anchors, per-residue tags, accept_same = False, threshold 10
    lchd = orc.LoCoHD(types, orc.WeightFunction(*bc.WF), orc.TagPairingRule({"accept_same": False}))

    def prims(tl):
        return [orc.PrimitiveAtom(pt, f"{fid[2]}/{fid[3][1]}-{rn}", c) for pt, c, (fid, rn, _nm) in tl]

    cent = [k for k, (pt, _c, _s) in enumerate(templates[0]) if pt == "Cent"]
    ref = prims(templates[0])
    per_frame = [lchd.from_primitives(ref, prims(tl), [(k, k) for k in cent], 10.0) for tl in templates[1:]]
    b_trj = np.zeros(len(templates[0]))
    b_trj[cent] = np.mean(per_frame, axis=0)
    (out / "blabelled_trajectory.pdb").write_text(aco.generate_primitive_pdb(types, templates[0], b_trj))
    # how far every label is from a rounding boundary of the %6.2f column (a HIP score 1e-15 away must print the same digits)
    for name, b in (("ensemble", b_ens), ("trajectory", b_trj)):
        frac = np.abs((np.asarray(b) * 100.0) % 1.0 - 0.5)
        print(f"{name}: {len(b)} primitive atoms, labels {np.min(b):.4f} .. {np.max(b):.4f}, closest to a rounding boundary: {frac.min() / 100:.2e}")


if __name__ == "__main__":
    main()
