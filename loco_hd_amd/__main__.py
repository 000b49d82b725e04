"""`python -m loco_hd_amd` -- the reference's command line over the MI355X scoring path.

Mirrors /root/reference/loco_hd/__main__.py: same flags (:49-134), same anchor-pairing file format
`chain/resnum-RESNAME/atom,atom:chain/resnum-RESNAME/atom;...` (:12-30, README.md:213-235), same output lines
`LoCoHD(<pair>) = <score>` (:203-204).  Structures are read with `loco_hd_amd.pdb_reader` (the reference uses
BioPython's PDBParser, which this image does not have); scoring runs on the GPU through LoCoHD.from_primitives.
"""
from __future__ import annotations

import json
import sys
from argparse import ArgumentParser, Namespace
from pathlib import Path
from typing import Dict, FrozenSet, List, Optional, Sequence, Tuple

from .api import LoCoHD, TagPairingRule, WeightFunction
from .atom_converter_utils import PrimitiveAssigner, PrimitiveAtomTemplate, prat_to_pra
from .pdb_reader import PDBParser

# chain ID (eg.: A), resi ID (eg.: 123-GLY), atom set (eg.: {CG, CZ})
TagIDType = Tuple[str, str, FrozenSet[str]]


def parse_anchor_pairing(anchor_pairing_str_list: Sequence[str]) -> List[Tuple[TagIDType, TagIDType]]:
    """__main__.py:12-30.  A malformed entry raises ValueError from the tuple unpacking, like the reference."""
    pairings = []
    for entry in anchor_pairing_str_list:
        tag1, tag2 = entry.split(":")
        chain1, resi1, atoms1 = tag1.split("/")
        chain2, resi2, atoms2 = tag2.split("/")
        pairings.append(((chain1, resi1, frozenset(atoms1.split(","))), (chain2, resi2, frozenset(atoms2.split(",")))))
    return pairings


def pra_template_list_to_idx_dict(pra_templates: Sequence[PrimitiveAtomTemplate]) -> Dict[TagIDType, int]:
    """__main__.py:33-47: (chain, "<resnum>-<resname>", atom set) -> index; a repeated id keeps its LAST index."""
    out: Dict[TagIDType, int] = {}
    for idx, prat in enumerate(pra_templates):
        src = prat.atom_source
        out[(src.source_residue[2], f"{src.source_residue[3][1]}-{src.source_residue_name}", frozenset(src.source_atom))] = idx
    return out


# (short flag, long flag, type, default, required, help) -- the flag names, types and defaults of __main__.py:49-134
_DEFAULT_TPRA = '{"accept_same": false}'
_DEFAULT_WFA = '{"function_name": "uniform", "parameters": [3.0, 10.0]}'
_FLAGS = (
    ("-s1", "--structure1", str, None, True, "first PDB file"),
    ("-s2", "--structure2", str, None, True, "second PDB file"),
    ("-pts", "--primitive_typing_scheme", str, None, True, "typing scheme (JSON: primitive type -> [[residue regex, atom regex, count]])"),
    ("-apf", "--anchor_pairing_file", Path, None, True,
     "text file with ';'-separated anchor pairs, each 'chain/resnum-RESNAME/atom,atom:chain/resnum-RESNAME/atom,...' "
     "(left side: primitive atom of structure 1, right side: of structure 2); line breaks are dropped, blanks are not"),
    ("-mn", "--model_number", int, 0, False, "model of both PDB files to use"),
    ("-nt", "--number_of_threads", int, None, False, "kept for compatibility: scoring runs on the GPU, there is no CPU thread pool"),
    ("-udc", "--upper_distance_cutoff", float, 10.0, False, "environment radius in angstrom"),
    ("-tpra", "--tag_pairing_rule_args", str, _DEFAULT_TPRA, False, f"JSON dict for TagPairingRule (default {_DEFAULT_TPRA})"),
    ("-wfa", "--weight_function_args", str, _DEFAULT_WFA, False, f"JSON dict for WeightFunction (default {_DEFAULT_WFA})"),
)


def parse_cli_args(argv: Optional[Sequence[str]] = None) -> Namespace:
    parser = ArgumentParser(prog="python -m loco_hd_amd", description="LoCoHD scores of anchor pairs of two PDB structures (MI355X)")
    for short, long_, kind, default, required, text in _FLAGS:
        parser.add_argument(short, long_, type=kind, default=default, required=required, help=text)
    ns = parser.parse_args(argv)
    for key in ("tag_pairing_rule_args", "weight_function_args"):
        setattr(ns, key, json.loads(getattr(ns, key)))
    return ns


def run(args: Namespace) -> List[str]:
    """__main__.py:149-204; returns the output lines."""
    with open(args.anchor_pairing_file, "r") as f:
        pair_strs = f.read().replace("\n", "").split(";")
    anchor_pairing = parse_anchor_pairing(pair_strs)

    structure1 = PDBParser(QUIET=True).get_structure("s1", args.structure1)[args.model_number]
    structure2 = PDBParser(QUIET=True).get_structure("s2", args.structure2)[args.model_number]
    assigner = PrimitiveAssigner(Path(args.primitive_typing_scheme))
    templates1 = assigner.assign_primitive_structure(structure1)
    templates2 = assigner.assign_primitive_structure(structure2)
    idx1, idx2 = pra_template_list_to_idx_dict(templates1), pra_template_list_to_idx_dict(templates2)
    anchor_pairs = [(idx1[a], idx2[b]) for a, b in anchor_pairing]  # unknown id -> KeyError, like the reference

    lchd = LoCoHD(assigner.all_primitive_types, WeightFunction(**args.weight_function_args),
                  TagPairingRule(args.tag_pairing_rule_args), args.number_of_threads)
    scores = lchd.from_primitives(list(map(prat_to_pra, templates1)), list(map(prat_to_pra, templates2)), anchor_pairs,
                                  args.upper_distance_cutoff)
    return [f"LoCoHD({s}) = {score}" for s, score in zip(pair_strs, scores)]


def main(argv: Optional[Sequence[str]] = None) -> int:
    for line in run(parse_cli_args(argv)):
        print(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
