"""Minimal PDB reader with the object model the reference's structure -> primitive-atom step walks.

The reference reads structures with BioPython (`Bio.PDB.PDBParser`, dependency `biopython>=1.80`,
/root/reference/pyproject.toml:18, tested with 1.81, README.md:50) and `PrimitiveAssigner` only ever touches
`structure.get_residues()`, `residue.resname`, `residue.full_id`, `residue.get_atoms()`, `atom.name`, `atom.coord`
(/root/reference/loco_hd/atom_converter_utils.py:95-118) plus `structure[model_number]`
(/root/reference/loco_hd/__main__.py:163-164).  BioPython is not part of this image, so this module restates
the documented behaviour of its permissive, QUIET parser for exactly that surface:

* ATOM / HETATM records of the coordinate section, fixed columns (name 13-16, altloc 17, resname 18-20 stripped,
  chain 22, resseq 23-26, icode 27, x/y/z 31-54, occupancy 55-60); coordinates are float32 like `Atom.coord`;
* MODEL / ENDMDL (models are numbered 0, 1, ... in file order; no MODEL record = one model 0); END / CONECT stop;
* residue id = (hetero flag, resseq, icode) with hetero flag " " (ATOM), "W" (HETATM HOH/WAT), "H_<resname>";
* a chain id that re-appears later in a model continues the existing chain; a residue id that re-appears with
  the same name continues the existing residue; with a different name (point mutation) the residue becomes
  disordered and the LAST added variant is the selected one; a hetero residue defined twice loses the atoms
  of the second definition (permissive mode drops the construction error);
* alternate locations: one entry per atom name, the variant with the highest occupancy is selected (first wins
  ties); an atom repeated with a blank altloc is dropped; names that collide only after stripping spaces keep
  their spaces.

Anything a LoCoHD workflow does not read (B factors, ANISOU, elements, segids, header) is not kept.
"""
from __future__ import annotations

import sys
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np

ResidueId = Tuple[str, int, str]


class PDBConstructionException(Exception):
    pass


class Atom:
    __slots__ = ("name", "fullname", "coord", "altloc", "occupancy", "serial_number", "parent", "index")

    def __init__(self, name: str, coord: np.ndarray, occupancy: Optional[float], altloc: str, fullname: str, serial_number: int):
        self.name, self.fullname, self.coord = name, fullname, coord
        self.altloc, self.occupancy, self.serial_number = altloc, occupancy, serial_number
        self.parent = None
        self.index = -1  # position in Structure.atom_table order (selected atoms only), set by Model.flatten()

    def get_name(self) -> str:
        return self.name

    def get_coord(self) -> np.ndarray:
        return self.coord

    @property
    def id(self) -> str:
        return self.name

    def __repr__(self) -> str:
        return f"<Atom {self.name}>"


class _AltlocSet:
    """All alternate locations of one atom name; attribute access goes to the selected one (DisorderedAtom)."""

    def __init__(self, name: str):
        self.id = name
        self.variants: Dict[str, Atom] = {}
        self.selected: Optional[Atom] = None
        self._best = -float(sys.maxsize)

    def add(self, atom: Atom) -> None:
        self.variants[atom.altloc] = atom
        occ = atom.occupancy
        if occ is not None and occ > self._best:
            self._best = occ
            self.selected = atom
        elif self.selected is None:
            self.selected = atom


class Residue:
    def __init__(self, res_id: ResidueId, resname: str):
        self.id, self.resname = res_id, resname
        self.parent: Optional["Chain"] = None
        self.full_id: Optional[tuple] = None
        self._entries: List[Union[Atom, _AltlocSet]] = []
        self._by_name: Dict[str, Union[Atom, _AltlocSet]] = {}

    def get_resname(self) -> str:
        return self.resname

    def get_id(self) -> ResidueId:
        return self.id

    def get_full_id(self) -> tuple:
        return self.full_id

    def get_atoms(self) -> Iterator[Atom]:
        for e in self._entries:
            yield e.selected if isinstance(e, _AltlocSet) else e

    __iter__ = get_atoms

    def __len__(self) -> int:
        return len(self._entries)

    def __getitem__(self, name: str) -> Atom:
        e = self._by_name[name]
        return e.selected if isinstance(e, _AltlocSet) else e

    def __contains__(self, name: str) -> bool:
        return name in self._by_name

    def __repr__(self) -> str:
        return f"<Residue {self.resname} het={self.id[0]} resseq={self.id[1]} icode={self.id[2]}>"

    # ---- construction (StructureBuilder.init_atom) ----------------------------------------------------------
    def _add_atom(self, name: str, fullname: str, coord, occupancy, altloc: str, serial: int) -> None:
        prev = self._by_name.get(name)
        if prev is not None:
            prev_full = prev.selected.fullname if isinstance(prev, _AltlocSet) else prev.fullname
            if prev_full != fullname:  # " CA " vs "CA  ": keep the spaces to tell them apart
                name = fullname
                prev = self._by_name.get(name)
        atom = Atom(name, coord, occupancy, altloc, fullname, serial)
        atom.parent = self
        if altloc != " ":
            if isinstance(prev, _AltlocSet):
                prev.add(atom)
            elif prev is not None:  # earlier copy had a blank altloc: both go into one altloc set, the new one first
                s = _AltlocSet(name)
                s.add(atom)
                s.add(prev)
                self._entries.remove(prev)  # detach_child + add: the altloc set takes the LAST position in the residue
                self._entries.append(s)
                self._by_name[name] = s
            else:
                s = _AltlocSet(name)
                s.add(atom)
                self._entries.append(s)
                self._by_name[name] = s
        else:
            if prev is not None:
                return  # "Atom defined twice": the construction error is dropped in permissive mode, the atom is lost
            self._entries.append(atom)
            self._by_name[name] = atom

    def _all_altloc(self) -> bool:
        return all(isinstance(e, _AltlocSet) for e in self._entries)


class _ResidueVariants:
    """Point mutation: several residues share one id; attribute access goes to the selected one (DisorderedResidue)."""

    def __init__(self, res_id: ResidueId):
        self.id = res_id
        self.variants: Dict[str, Residue] = {}
        self.selected: Optional[Residue] = None

    def add(self, residue: Residue) -> None:
        self.variants[residue.resname] = residue
        self.selected = residue


class Chain:
    def __init__(self, chain_id: str):
        self.id = chain_id
        self.parent: Optional["Model"] = None
        self._entries: List[Union[Residue, _ResidueVariants]] = []
        self._by_id: Dict[ResidueId, Union[Residue, _ResidueVariants]] = {}

    def get_id(self) -> str:
        return self.id

    def get_residues(self) -> Iterator[Residue]:
        for e in self._entries:
            yield e.selected if isinstance(e, _ResidueVariants) else e

    __iter__ = get_residues

    def __len__(self) -> int:
        return len(self._entries)

    def __getitem__(self, res_id) -> Residue:
        if isinstance(res_id, int):
            res_id = (" ", res_id, " ")
        e = self._by_id[res_id]
        return e.selected if isinstance(e, _ResidueVariants) else e

    def get_atoms(self) -> Iterator[Atom]:
        for r in self.get_residues():
            yield from r.get_atoms()

    # ---- construction (StructureBuilder.init_residue) --------------------------------------------------------
    def _open_residue(self, resname: str, hetero_flag: str, resseq: int, icode: str) -> Optional[Residue]:
        field = "H_" + resname if hetero_flag == "H" else hetero_flag
        res_id = (field, resseq, icode)
        prev = self._by_id.get(res_id)
        if prev is not None:
            if field != " ":
                return None  # "defined twice": dropped in permissive mode, the atoms that follow go nowhere
            if isinstance(prev, _ResidueVariants):
                if resname in prev.variants:
                    prev.selected = prev.variants[resname]
                    return prev.selected
                new = self._new_residue(res_id, resname)
                prev.add(new)
                return new
            if prev.resname == resname:
                return prev
            if not prev._all_altloc():
                return None  # "Blank altlocs in duplicate residue": the second definition is lost
            var = _ResidueVariants(res_id)
            var.add(prev)
            new = self._new_residue(res_id, resname)
            var.add(new)
            self._entries.remove(prev)
            self._entries.append(var)
            self._by_id[res_id] = var
            return new
        new = self._new_residue(res_id, resname)
        self._entries.append(new)
        self._by_id[res_id] = new
        return new

    def _new_residue(self, res_id: ResidueId, resname: str) -> Residue:
        r = Residue(res_id, resname)
        r.parent = self
        m = self.parent
        r.full_id = (m.parent.id, m.id, self.id, res_id)
        return r


class Model:
    def __init__(self, model_id: int, serial_num: Optional[int] = None):
        self.id = model_id
        self.serial_num = model_id if serial_num is None else serial_num
        self.parent: Optional["Structure"] = None
        self._chains: List[Chain] = []
        self._by_id: Dict[str, Chain] = {}

    def get_chains(self) -> Iterator[Chain]:
        return iter(self._chains)

    __iter__ = get_chains

    def __len__(self) -> int:
        return len(self._chains)

    def __getitem__(self, chain_id: str) -> Chain:
        return self._by_id[chain_id]

    def get_residues(self) -> Iterator[Residue]:
        for c in self._chains:
            yield from c.get_residues()

    def get_atoms(self) -> Iterator[Atom]:
        for c in self._chains:
            yield from c.get_atoms()

    def _open_chain(self, chain_id: str) -> Chain:
        c = self._by_id.get(chain_id)
        if c is None:  # a chain id seen before is "discontinuous": the existing chain continues
            c = Chain(chain_id)
            c.parent = self
            self._chains.append(c)
            self._by_id[chain_id] = c
        return c

    def coordinates(self) -> np.ndarray:
        """float32 [n_atoms][3] of the selected atoms in get_atoms() order; also numbers them (Atom.index)."""
        atoms = list(self.get_atoms())
        for i, a in enumerate(atoms):
            a.index = i
        return np.stack([a.coord for a in atoms]).astype(np.float32) if atoms else np.zeros((0, 3), np.float32)


class Structure:
    def __init__(self, structure_id: str):
        self.id = structure_id
        self._models: List[Model] = []

    def get_models(self) -> Iterator[Model]:
        return iter(self._models)

    __iter__ = get_models

    def __len__(self) -> int:
        return len(self._models)

    def __getitem__(self, model_id: int) -> Model:
        for m in self._models:
            if m.id == model_id:
                return m
        raise KeyError(model_id)

    def get_chains(self) -> Iterator[Chain]:
        for m in self._models:
            yield from m.get_chains()

    def get_residues(self) -> Iterator[Residue]:
        for m in self._models:
            yield from m.get_residues()

    def get_atoms(self) -> Iterator[Atom]:
        for m in self._models:
            yield from m.get_atoms()


class PDBParser:
    """`PDBParser(QUIET=True).get_structure(id, file)` as used in the reference (README.md:255-259, __main__.py:163)."""

    def __init__(self, PERMISSIVE: bool = True, QUIET: bool = True):
        self.permissive, self.quiet = bool(PERMISSIVE), bool(QUIET)

    def get_structure(self, structure_id: str, file) -> Structure:
        if hasattr(file, "read"):
            lines = file.read().splitlines()
        else:
            lines = Path(file).read_text().splitlines()
        return self.parse_lines(structure_id, lines)

    def parse_lines(self, structure_id: str, lines) -> Structure:
        st = Structure(structure_id)
        # the header is everything before the first coordinate record
        start = 0
        for start, line in enumerate(lines):
            if line[0:6] in ("ATOM  ", "HETATM", "MODEL "):
                break
        else:
            return st
        model: Optional[Model] = None
        next_model_id = 0
        chain: Optional[Chain] = None
        residue: Optional[Residue] = None
        cur_chain_id = cur_res_id = cur_resname = None
        for lineno in range(start, len(lines)):
            line = lines[lineno].rstrip("\n")
            rec = line[0:6]
            if not line.strip():
                continue
            if rec == "ATOM  " or rec == "HETATM":
                if model is None:
                    model = self._new_model(st, next_model_id, None)
                    next_model_id += 1
                fullname = line[12:16]
                parts = fullname.split()
                name = parts[0] if len(parts) == 1 else fullname
                altloc = line[16:17] or " "
                resname = line[17:20].strip()
                chain_id = line[21:22] or " "
                try:
                    serial = int(line[6:11])
                except ValueError:
                    serial = 0
                try:
                    resseq = int(line[22:26].split()[0])
                except (ValueError, IndexError):
                    raise PDBConstructionException(f"Invalid residue number at line {lineno + 1}.") from None
                icode = line[26:27] or " "
                hetero = " " if rec == "ATOM  " else ("W" if resname in ("HOH", "WAT") else "H")
                res_id = (hetero, resseq, icode)
                try:
                    coord = np.array((float(line[30:38]), float(line[38:46]), float(line[46:54])), "f")
                except ValueError:
                    raise PDBConstructionException(f"Invalid or missing coordinate(s) at line {lineno + 1}.") from None
                try:
                    occupancy: Optional[float] = float(line[54:60])
                except ValueError:
                    occupancy = None
                if cur_chain_id != chain_id:
                    cur_chain_id = chain_id
                    chain = model._open_chain(chain_id)
                    cur_res_id, cur_resname = res_id, resname
                    residue = chain._open_residue(resname, hetero, resseq, icode)
                elif cur_res_id != res_id or cur_resname != resname:
                    cur_res_id, cur_resname = res_id, resname
                    residue = chain._open_residue(resname, hetero, resseq, icode)
                if residue is not None:
                    residue._add_atom(name, fullname, coord, occupancy, altloc, serial)
            elif rec == "MODEL ":
                try:
                    serial_num = int(line[10:14])
                except ValueError:
                    serial_num = 0
                model = self._new_model(st, next_model_id, serial_num)
                next_model_id += 1
                cur_chain_id = cur_res_id = None
            elif rec == "END   " or rec == "CONECT":
                break
            elif rec == "ENDMDL":
                model = None
                cur_chain_id = cur_res_id = None
        return st

    @staticmethod
    def _new_model(st: Structure, model_id: int, serial_num: Optional[int]) -> Model:
        m = Model(model_id, serial_num)
        m.parent = st
        st._models.append(m)
        return m
