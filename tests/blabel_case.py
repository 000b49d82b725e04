"""The two B-factor-labelled primitive-structure cases of tests/golden/ (row f4 of SURVEY.md section 8).

Callers of the reference put per-atom LoCoHD scores into the B column of a primitive-atom PDB file
(/root/reference/loco_hd/atom_converter_utils.py:133-168, called at python_codes/ensembles/compare_ensembles.py:323 and
python_codes/analyze_singles.py:149-153).  This module defines the INPUTS of the two cases -- seeded synthetic PDB texts, nothing
else -- and the two caller recipes as functions of a backend module `mod` (the CPU oracle for tests/golden/make_blabelled_pdb.py,
the product package through its `loco_hd` import name for tests/test_gpu_blabels.py):

  ensemble    compare_ensembles.py:196-323: K conformers, distance matrices with homo-residue contacts banned (+inf), from_dmxs
              over every pair of conformers, mean score per primitive atom -> B labels of conformer 0
  trajectory  trajectory_analyzer.py:97-120 (BASELINE config 4's pipeline): conformer 0 against the others as frames, "Cent"
              anchors, accept_same=False, uniform[3,10], threshold 10 -> mean per anchor over the frames; non-anchor atoms 0.0
"""
from __future__ import annotations

import numpy as np

import pdb_util

N_CONFORMERS = 5
WF = ("uniform", [3.0, 10.0])


def conformer_texts():
    """K PDB texts: a seeded synthetic structure and K-1 copies whose atoms moved by N(0, 0.6 A); a member IS its text (the
    coordinates are what the %8.3f columns hold, for every reader alike)."""
    base = pdb_util.synthetic_pdb(seed=51, n_res=24, chains="A", box=24.0, altlocs=False, hetero=False, insertions=False)
    rng = np.random.default_rng(52)
    texts = [base]
    for _ in range(N_CONFORMERS - 1):
        out = []
        for line in base.splitlines():
            if line.startswith(("ATOM", "HETATM")):
                xyz = np.array([float(line[30:38]), float(line[38:46]), float(line[46:54])]) + rng.normal(0.0, 0.6, 3)
                line = f"{line[:30]}{xyz[0]:8.3f}{xyz[1]:8.3f}{xyz[2]:8.3f}{line[54:]}"
            out.append(line)
        texts.append("\n".join(out) + "\n")
    return texts


def ensemble_b_labels(mod, types, template_lists):
    """compare_ensembles.py:219-300 on lists of templates given as (primitive_type, coordinates, residue_id) triples."""
    seq = [t for t, _c, _r in template_lists[0]]
    n = len(seq)
    homo = [(i, j) for i in range(n) for j in range(n) if i != j and template_lists[0][i][2] == template_lists[0][j][2]]
    dmxs = []
    for tl in template_lists:
        c = np.array([np.asarray(co, dtype=np.float64) for _t, co, _r in tl])
        d = c[np.newaxis, ...] - c[:, np.newaxis, :]
        d = np.sqrt(np.sum(d ** 2, axis=2))
        for i, j in homo:
            d[i][j] = float("inf")
        dmxs.append(d)
    lchd = mod.LoCoHD(types, mod.WeightFunction(*WF))
    by_atom = []
    for i in range(len(dmxs)):
        for j in range(i + 1, len(dmxs)):
            by_atom.append(lchd.from_dmxs(seq, seq, dmxs[i], dmxs[j]))
    return np.mean(by_atom, axis=0)
