"""Structure -> primitive-atom conversion: the step in front of the scoring path.

Mirrors /root/reference/loco_hd/atom_converter_utils.py (same class names, fields and results):

    PrimitiveAtomSource, PrimitiveAtomTemplate, TypingSchemeElement        :19-57
    PrimitiveAssigner.__init__ (typing-scheme JSON), all_primitive_types   :66-91
    PrimitiveAssigner.assign_primitive_structure                           :93-131
    PrimitiveAssigner.generate_primitive_pdb                               :133-168

The structure argument is duck-typed exactly as far as the reference walks it (`get_residues()`, `resname`,
`full_id`, `get_atoms()`, `name`, `coord`): BioPython entities work, and so do the entities of
`loco_hd_amd.pdb_reader` (BioPython is not in this image).

MI355X-first addition (`compile_topology`): which atoms feed which primitive atom depends only on residue and
atom NAMES, i.e. on the topology, not on the coordinates.  `compile_topology` resolves the regular expressions
once into a CSR index table (`PrimitiveTopology`); from then on a frame of an MD trajectory is converted by the
`k_frames_centroids` kernel on the device (DeviceSession.score_trajectory(..., topology=...)) instead of the
per-frame Python loop of python_codes/trajectory_analyzer.py:37-74, with np.mean's float32 arithmetic bit for bit.
"""
from __future__ import annotations

import json
import re
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .api import LoCoHD, PrimitiveAtom

ResiFullIdType = Tuple[str, int, str, Tuple[str, int, str]]  # structure id, model id, chain id, residue id


@dataclass
class PrimitiveAtomSource:
    """atom_converter_utils.py:19-31: the residue a primitive atom comes from and the atom names it was built from
    (one name = a plain atom, several = a centroid)."""

    source_residue: ResiFullIdType
    source_residue_name: str
    source_atom: List[str]


@dataclass
class PrimitiveAtomTemplate:
    """atom_converter_utils.py:34-43: intermediate between a structure's atoms and a `PrimitiveAtom`."""

    primitive_type: str
    coordinates: np.ndarray
    atom_source: PrimitiveAtomSource


@dataclass
class TypingSchemeElement:
    """atom_converter_utils.py:46-57."""

    primitive_type: str
    residue_matcher: "re.Pattern[str]"
    atom_matcher: "re.Pattern[str]"
    atom_counter: Union[int, str]

    def match_resi(self, resi_name: str) -> bool:
        return self.residue_matcher.fullmatch(resi_name) is not None

    def match_atom(self, atom_name: str) -> bool:
        return self.atom_matcher.fullmatch(atom_name) is not None


def prat_to_pra(prat: PrimitiveAtomTemplate) -> PrimitiveAtom:
    """The template -> PrimitiveAtom conversion every reference workflow uses (README.md:316-328,
    loco_hd/__main__.py:136-146): tag = "<chain id>/<residue number>-<residue name>"."""
    resi_id = prat.atom_source.source_residue
    return PrimitiveAtom(prat.primitive_type, f"{resi_id[2]}/{resi_id[3][1]}-{prat.atom_source.source_residue_name}",
                         prat.coordinates)


class PrimitiveTopology:
    """CSR form of one structure's primitive atoms: primitive atom p is the float32 centroid of source atoms
    `src_idx[src_start[p]:src_start[p+1]]`, positions in the `get_atoms()` order of the structure it was compiled
    from.  `primitive_types[p]`, `sources[p]` and `tags[p]` (the prat_to_pra tag) describe it."""

    def __init__(self, primitive_types: List[str], sources: List[PrimitiveAtomSource], src_start: np.ndarray, src_idx: np.ndarray,
                 n_atoms: int, atom_coords: np.ndarray):
        self.primitive_types, self.sources = primitive_types, sources
        self.src_start, self.src_idx, self.n_atoms = src_start, src_idx, int(n_atoms)
        self.atom_coords = atom_coords  # float32 [n_atoms][3] of the structure it was compiled from
        self.tags = [f"{s.source_residue[2]}/{s.source_residue[3][1]}-{s.source_residue_name}" for s in sources]

    def __len__(self) -> int:
        return len(self.primitive_types)

    def centroids(self, atom_coords: Optional[np.ndarray] = None) -> np.ndarray:
        """float32 [..., n_primitive][3]: np.mean(members, axis=0) for every primitive atom (sequential float32 adds in
        member order, then one division -- the arithmetic of atom_converter_utils.py:126).  Host-side helper for single
        structures; frames of a trajectory are converted on the device."""
        xyz = self.atom_coords if atom_coords is None else np.asarray(atom_coords, dtype=np.float32)
        if xyz.shape[-2:] != (self.n_atoms, 3):
            raise ValueError(f"expected [..., {self.n_atoms}, 3] source-atom coordinates, got {xyz.shape}")
        lens = np.diff(self.src_start)
        if len(lens) and lens.min() < 1:
            raise ValueError("a primitive atom without source atoms has no centroid")
        acc = np.zeros(xyz.shape[:-2] + (len(lens), 3), dtype=np.float32)  # add.reduce starts from +0 (a lone -0.0 -> +0.0)
        for k in range(int(lens.max()) if len(lens) else 0):
            sel = np.nonzero(lens > k)[0]
            acc[..., sel, :] += xyz[..., self.src_idx[self.src_start[sel] + k], :]
        return (acc / lens.astype(np.float32)[:, None]).astype(np.float32, copy=False)

    def templates(self, atom_coords: Optional[np.ndarray] = None) -> List[PrimitiveAtomTemplate]:
        cen = self.centroids(atom_coords)
        return [PrimitiveAtomTemplate(t, cen[p], s) for p, (t, s) in enumerate(zip(self.primitive_types, self.sources))]

    def pack(self, lchd: LoCoHD, interner: Optional[Dict[str, int]] = None, atom_coords: Optional[np.ndarray] = None):
        """SoA arrays (xyz f64 [n][3], category i32, tag i32) for `LoCoHD.from_packed` / `DeviceSession.upload` without
        a list of PrimitiveAtom objects in between."""
        from .api import _Packed

        interner = {} if interner is None else interner
        cat = lchd._cats(self.primitive_types)
        tag = np.fromiter((interner.setdefault(t, len(interner)) for t in self.tags), dtype=np.int32, count=len(self.tags))
        return _Packed(self.centroids(atom_coords).astype(np.float64), cat, tag)


class PrimitiveAssigner:
    """atom_converter_utils.py:60-168.  `config_path` is a typing-scheme JSON:
    {primitive type: [[residue regex, atom regex(, atom count | "any")], ...]}."""

    def __init__(self, config_path: Union[str, Path]):
        with open(config_path, "r") as f:
            config: Dict[str, List[Sequence[Any]]] = json.load(f)
        self.scheme: List[TypingSchemeElement] = []
        for primitive_type, scheme_elements in config.items():
            for element in scheme_elements:
                counter = 1 if len(element) == 2 else element[2]
                self.scheme.append(TypingSchemeElement(primitive_type, re.compile(element[0]), re.compile(element[1]), counter))
        self._resi_cache: Dict[str, List[int]] = {}
        self._atom_cache: Dict[Tuple[int, str], bool] = {}

    @property
    def all_primitive_types(self) -> List[str]:
        """The reference returns `list({...})` (:85-87), i.e. the types in a per-process arbitrary order; here they come in
        the order of the typing scheme, which is one of those orders and the same in every process / on every rank."""
        return list(dict.fromkeys(e.primitive_type for e in self.scheme))

    @all_primitive_types.setter
    def all_primitive_types(self, value):
        raise Exception("Cannot set all_primitive_types directly, since it depends on the config file!")

    # ---- name matching with memoised regular expressions -----------------------------------------------------------
    def _elements_for(self, resi_name: str) -> List[int]:
        hit = self._resi_cache.get(resi_name)
        if hit is None:
            hit = self._resi_cache[resi_name] = [i for i, e in enumerate(self.scheme) if e.match_resi(resi_name)]
        return hit

    def _atom_ok(self, element: int, atom_name: str) -> bool:
        key = (element, atom_name)
        hit = self._atom_cache.get(key)
        if hit is None:
            hit = self._atom_cache[key] = self.scheme[element].match_atom(atom_name)
        return hit

    def _walk(self, structure):
        """Yield (typing-scheme element, residue, matched atoms) for every primitive atom, in the reference's order:
        residues as get_residues() gives them, scheme elements in file order, atoms in residue order (:95-129)."""
        for resi in structure.get_residues():
            elements = self._elements_for(resi.resname)
            if not elements:
                continue
            atoms = list(resi.get_atoms())
            names = [a.name for a in atoms]
            for ei in elements:
                tse = self.scheme[ei]
                members = [a for a, nm in zip(atoms, names) if self._atom_ok(ei, nm)]
                if tse.atom_counter == "any":
                    pass
                elif tse.atom_counter == len(members):
                    pass
                else:
                    continue
                yield tse, resi, members

    def assign_primitive_structure(self, structure) -> List[PrimitiveAtomTemplate]:
        out = []
        for tse, resi, members in self._walk(structure):
            centroid = np.mean([a.coord for a in members], axis=0)
            out.append(PrimitiveAtomTemplate(tse.primitive_type, centroid,
                                             PrimitiveAtomSource(resi.full_id, resi.resname, [a.name for a in members])))
        return out

    def compile_topology(self, structure) -> PrimitiveTopology:
        """Resolve the typing scheme against `structure` once: a CSR table primitive atom -> source atoms."""
        order = {id(a): i for i, a in enumerate(structure.get_atoms())}
        coords = np.zeros((len(order), 3), dtype=np.float32)
        for a in structure.get_atoms():
            coords[order[id(a)]] = a.coord
        types, sources, start, idx = [], [], [0], []
        for tse, resi, members in self._walk(structure):
            if not members:
                raise ValueError(f"typing-scheme element {tse.primitive_type!r} matched residue {resi.full_id} without any atom: "
                                 "its centroid is undefined (np.mean of an empty list)")
            types.append(tse.primitive_type)
            sources.append(PrimitiveAtomSource(resi.full_id, resi.resname, [a.name for a in members]))
            idx.extend(order[id(a)] for a in members)
            start.append(len(idx))
        return PrimitiveTopology(types, sources, np.asarray(start, dtype=np.int32), np.asarray(idx, dtype=np.int32), len(order), coords)

    def generate_primitive_pdb(self, primitive_structure: List[PrimitiveAtomTemplate],
                               b_labels: Union[None, List[float], np.ndarray] = None) -> str:
        """atom_converter_utils.py:133-168, column for column -- including that column 22 receives element [1] of the
        residue's full id (the model number), which is what the reference writes there."""
        types = self.all_primitive_types
        lines = []
        last_resi_id = None
        resi_idx = 0
        for i, prat in enumerate(primitive_structure):
            resi_id = prat.atom_source.source_residue
            b_factor = 1.0 if b_labels is None else b_labels[i]
            if resi_id != last_resi_id:
                resi_idx += 1
                last_resi_id = resi_id
            atom_name = chr(65 + types.index(prat.primitive_type))
            c = prat.coordinates
            lines.append(f"ATOM  {i + 1: >5} {atom_name: >4} {prat.atom_source.source_residue_name} {resi_id[1]}{resi_idx: >4}    "
                         f"{c[0]:8.3f}{c[1]:8.3f}{c[2]:8.3f}{1.:6.2f}{b_factor:6.2f}          Pr  \n")
        return "".join(lines)
