"""The reference's Python surface (loco_hd/loco_hd.pyi) over the MI355X-native core.

Five classes with the reference's names, argument meaning and error behaviour:

    WeightFunction        /root/reference/src/locohd/weight_function.rs:6-120
    PrimitiveAtom         /root/reference/src/locohd/primitive_atom.rs:4-25
    TagPairingRule        /root/reference/src/locohd/tag_pairing_rule.rs:5-75
    StatisticalDistance   /root/reference/src/locohd/pmf/statistical_distances.rs:87-142
    LoCoHD                /root/reference/src/locohd.rs:42-55, 286-568

This module only validates, interns strings (categories -> the index the reference's HashMap gives
them, tags -> integers) and packs contiguous NumPy buffers; all scoring happens in
libloco_hd_hip.so on the GPU.  There is no CPU fallback: without a usable MI355X the four from_*
methods raise DeviceError.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _native as N

try:  # built by `make -C loco_hd_amd/csrc`; argument conversion only -- without it the same conversion runs as a Python loop
    from . import _fastpack
except ImportError:  # pragma: no cover
    _fastpack = None

WF_KINDS = {"hyper_exp": 0, "dagum": 1, "uniform": 2, "kumaraswamy": 3}
SD_KINDS = {"Hellinger": 0, "Kolmogorov-Smirnov": 1, "Kullback-Leibler": 2, "Renyi": 3}


def _f64(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.float64)


class WeightFunction:
    """weight_function.rs:6-120.  ``parameters`` and ``function_name`` are read-only like the pyo3 getters."""

    __slots__ = ("_name", "_params", "_kind", "_p")

    def __init__(self, function_name: str, parameters: Sequence[float]) -> None:
        name = str(function_name)
        self._params = [float(x) for x in parameters]
        self._p = _f64(self._params)
        if name not in WF_KINDS:  # weight_function.rs:85-89
            raise ValueError(f'No function implemented with name "{name}"!')
        self._name, self._kind = name, WF_KINDS[name]
        N.check(N.lib().lchd_wf_validate(self._kind, N.dp(self._p), len(self._p)))

    @property
    def parameters(self) -> List[float]:
        return list(self._params)

    @property
    def function_name(self) -> str:
        return self._name

    def integral_vec(self, points: Sequence[float]) -> List[float]:
        x = _f64(points).reshape(-1)
        out = np.empty_like(x)
        N.check(N.lib().lchd_wf_cdf(self._kind, N.dp(self._p), len(self._p), N.dp(x), x.size, N.dp(out)))
        return out.tolist()

    def integral_point(self, point: float) -> float:
        return self.integral_vec([float(point)])[0]

    def integral_range(self, point_from: float, point_to: float) -> float:
        hi = self.integral_point(point_to)  # weight_function.rs:118-120 evaluates the upper bound first
        return hi - self.integral_point(point_from)

    def _c(self) -> N.WeightFunctionC:
        return N.WeightFunctionC(self._kind, len(self._p), N.dp(self._p))

    def __repr__(self) -> str:
        return f"WeightFunction({self._name!r}, {self._params!r})"


# Bumped by every PrimitiveAtom SETTER (not by construction): LoCoHD caches the packed form of the lists it is given and a
# cached list is only valid while none of its atoms has been changed (see LoCoHD._packed_lists).
_ATOM_MUTATIONS = [0]


# Process-wide string tables of PrimitiveAtom: every atom interns its two strings ONCE, when it is constructed or changed, and carries
# the ids (slots _pid / _tid) -- packing a list for a from_primitives call is then a gather of three doubles and two integers per
# atom in native code (_fastpack.pack_atoms), whatever list object the atoms arrive in.  The reference's PyO3 extraction clones
# both Strings of every atom on every call instead (primitive_atom.rs:4-16, src/locohd.rs:479-485).  Ids are never re-used; the
# tables grow with the number of DISTINCT strings the process has seen (residue tags, primitive type names).
_TYPE_IDS: Dict[Any, int] = {}
_TYPE_NAMES: List[Any] = []
_TAG_IDS: Dict[Any, int] = {}


def _type_id(value) -> int:
    try:
        i = _TYPE_IDS.get(value)
    except TypeError:  # unhashable: no id (the general packing path looks str(value) up)
        return -1
    if i is None:
        i = _TYPE_IDS[value] = len(_TYPE_NAMES)
        _TYPE_NAMES.append(value)
    return i


def _tag_id(value) -> int:
    try:
        return _TAG_IDS.setdefault(value, len(_TAG_IDS))
    except TypeError:
        return -1


class PrimitiveAtom:
    """primitive_atom.rs:4-25: a record with get+set attributes."""

    __slots__ = ("_primitive_type", "_tag", "_coordinates", "_pid", "_tid")

    def __init__(self, primitive_type: str, tag: str, coordinates: Sequence[float]) -> None:
        self._primitive_type = primitive_type
        self._tag = tag
        v = [float(x) for x in coordinates]
        if len(v) != 3:
            raise ValueError(f"expected a sequence of length 3 (got {len(v)})")
        self._coordinates = v
        self._pid = _type_id(primitive_type)
        self._tid = _tag_id(tag)

    @property
    def primitive_type(self) -> str:
        return self._primitive_type

    @primitive_type.setter
    def primitive_type(self, value: str) -> None:
        _ATOM_MUTATIONS[0] += 1
        self._primitive_type = value
        self._pid = _type_id(value)

    @property
    def tag(self) -> str:
        return self._tag

    @tag.setter
    def tag(self, value: str) -> None:
        _ATOM_MUTATIONS[0] += 1
        self._tag = value
        self._tid = _tag_id(value)

    @property
    def coordinates(self) -> List[float]:
        return list(self._coordinates)

    @coordinates.setter
    def coordinates(self, value: Sequence[float]) -> None:
        v = [float(x) for x in value]
        if len(v) != 3:
            raise ValueError(f"expected a sequence of length 3 (got {len(v)})")
        _ATOM_MUTATIONS[0] += 1
        self._coordinates = v

    def __getstate__(self):
        return (self._primitive_type, self._tag, self._coordinates)

    def __setstate__(self, state):
        self._primitive_type, self._tag, self._coordinates = state
        self._pid, self._tid = _type_id(self._primitive_type), _tag_id(self._tag)

    def __repr__(self) -> str:
        return f"PrimitiveAtom({self._primitive_type!r}, {self._tag!r}, {self._coordinates!r})"


class TagPairingRule:
    """tag_pairing_rule.rs:5-75.  The dict is tried as {"accept_same"} first, then as
    {"tag_pairs", "accepted_pairs", "ordered"} (derive(FromPyObject) order)."""

    def __init__(self, variant: Dict[str, Any]) -> None:
        if not hasattr(variant, "__getitem__"):
            raise TypeError("TagPairingRule expects a dict")
        if "accept_same" in variant:
            self._mode, self._accept_same = 0, bool(variant["accept_same"])
            self._pairs, self._accepted_pairs, self._ordered = frozenset(), True, True
        elif all(k in variant for k in ("tag_pairs", "accepted_pairs", "ordered")):
            self._mode, self._accept_same = 1, True
            self._pairs = frozenset((str(a), str(b)) for a, b in variant["tag_pairs"])
            self._accepted_pairs, self._ordered = bool(variant["accepted_pairs"]), bool(variant["ordered"])
        else:
            raise TypeError("failed to extract enum TagPairingRuleVariants ('WithoutList | WithList')")

    def pair_accepted(self, pair: Tuple[str, str]) -> bool:
        t0, t1 = pair
        if self._mode == 0:  # :53-61
            accepted = t0 == t1
            return accepted if self._accept_same else not accepted
        accepted = (t0, t1) in self._pairs  # :63-74
        if not self._ordered:
            accepted = accepted or (t1, t0) in self._pairs
        return accepted if self._accepted_pairs else not accepted

    def get_dbg_str(self) -> str:
        if self._mode == 0:
            return f"TagPairingRule {{\n    variant: WithoutList {{\n        accept_same: {str(self._accept_same).lower()},\n    }},\n}}"
        pairs = "".join(f'            (\n                "{a}",\n                "{b}",\n            ),\n' for a, b in sorted(self._pairs))
        return ("TagPairingRule {\n    variant: WithList {\n        tag_pairs: {\n" + pairs + "        },\n"
                f"        accepted_pairs: {str(self._accepted_pairs).lower()},\n        ordered: {str(self._ordered).lower()},\n    }},\n}}")


class StatisticalDistance:
    """statistical_distances.rs:87-142."""

    def __init__(self, distance_name: str, parameters: Sequence[float]) -> None:
        name = str(distance_name)
        self._params = [float(x) for x in parameters]
        if name not in SD_KINDS:  # :112-115
            raise ValueError(f"Invalid statistical distance name {name}!")
        self._name, self._kind = name, SD_KINDS[name]
        N.check(N.lib().lchd_sd_validate(self._kind, len(self._params)))

    def run(self, p1: Sequence[float], p2: Sequence[float]) -> float:
        a, b = _f64(p1).reshape(-1), _f64(p2).reshape(-1)
        n = min(a.size, b.size)  # iter().zip() stops at the shorter one
        prm = np.zeros(2)
        prm[: len(self._params)] = self._params
        out = C.c_double()
        N.check(N.lib().lchd_sd_run(self._kind, N.dp(prm), N.dp(a), N.dp(b), n, C.cast(C.byref(out), N._DP)))
        return out.value

    def __repr__(self) -> str:
        return f"StatisticalDistance({self._name!r}, {self._params!r})"


class _Packed:
    """A structure interned into SoA arrays (xyz [n][3] f64, category i32, tag i32)."""

    __slots__ = ("xyz", "cat", "tag")

    def __init__(self, xyz, cat, tag):
        self.xyz, self.cat, self.tag = xyz, cat, tag


class LoCoHD:
    """src/locohd.rs:42-55, 286-568.

    Same constructor as the reference; ``n_of_threads`` is accepted and ignored (the rayon pool is
    replaced by the GPU).  ``device`` (keyword-only, additive) selects the HIP device; default = current.
    ``devices`` (keyword-only, additive): a list of HIP device ordinals -- the counterpart of the reference's thread pool,
    which is a field of the instance (src/locohd.rs:53,373-383): ``from_primitives`` then spreads the anchor pairs of every
    call over these GPUs inside the C library (lchd_group_from_primitives), results in anchor-pair order as always.
    ``deterministic`` (keyword-only, additive): pin ONE sweep kernel family (lchd_ctx_set_deterministic) -- like the reference's
    single code path (src/locohd.rs:61-226) a pair's score is then bitwise independent of the other pairs of the call and of what
    the instance scored before, at about half the default throughput; the default picks kernels per call (<= 1e-13 apart).
    """

    def __init__(self, categories: Sequence[str], w_func: Union[None, WeightFunction, Dict[str, WeightFunction]] = None,
                 tag_pairing_rule: Optional[TagPairingRule] = None, n_of_threads: Optional[int] = None,
                 category_weights: Optional[Sequence[float]] = None,
                 statistical_distance: Optional[StatisticalDistance] = None, *, device: Optional[int] = None,
                 devices: Optional[Sequence[int]] = None, deterministic: bool = False) -> None:
        names = [str(c) for c in categories]
        cat_map: Dict[str, int] = {}
        for i, nm in enumerate(names):  # :312-316 HashMap collect: a repeated name keeps its last index
            cat_map[nm] = i
        w = np.ones(len(cat_map)) if category_weights is None else _f64(list(category_weights)).reshape(-1)
        N.check(N.lib().lchd_config_validate(len(names), len(cat_map), N.dp(w), w.size))  # :305-346
        if n_of_threads is not None and int(n_of_threads) < 0:
            raise OverflowError("can't convert negative int to unsigned")
        self._categories, self._weights = cat_map, w
        if w_func is None:  # :349-354
            w_func = WeightFunction("uniform", [3.0, 10.0])
        elif not isinstance(w_func, WeightFunction):
            if not isinstance(w_func, dict) or not all(isinstance(v, WeightFunction) for v in w_func.values()):
                raise TypeError("w_func must be None, a WeightFunction or a dict of WeightFunctions")
            w_func = {str(k): v for k, v in w_func.items()}
        self._w_func = w_func
        self._tpr = TagPairingRule({"accept_same": True}) if tag_pairing_rule is None else tag_pairing_rule  # :357-362
        self._sd = StatisticalDistance("Hellinger", [2.0]) if statistical_distance is None else statistical_distance  # :365-370
        if not isinstance(self._tpr, TagPairingRule):
            raise TypeError("tag_pairing_rule must be a TagPairingRule")
        if not isinstance(self._sd, StatisticalDistance):
            raise TypeError("statistical_distance must be a StatisticalDistance")
        self._n_threads = n_of_threads
        self._device = -1 if device is None else int(device)
        self._devices = None if devices is None else [int(d) for d in devices]
        if self._devices is not None and not self._devices:
            raise ValueError("devices must name at least one HIP device")
        if self._devices is not None and device is None:
            self._device = self._devices[0]  # the single-device entry points (from_anchors, from_dmxs, from_coords)
        self._deterministic = bool(deterministic)
        if self._deterministic and self._devices is not None and len(self._devices) > 1:
            raise ValueError("deterministic=True applies to one device (the device group picks kernels per share)")
        self._ctx = None
        self._group = None
        self._pack_cache: List[Any] = []   # most recent first: (list, its items, mutation stamp, None, packed, interner)
        self._pid_map = np.empty(0, dtype=np.int32)  # PrimitiveAtom type id -> category index (_type_map)
        self._anchor_cache = None
        self._wf_names = list(self._w_func) if isinstance(self._w_func, dict) else None

    # ---- getters (#[pyo3(get)], :45-52) -------------------------------------------------------------
    @property
    def categories(self) -> Dict[str, int]:
        return dict(self._categories)

    @property
    def category_weights(self) -> List[float]:
        return self._weights.tolist()

    @property
    def w_func(self):
        return self._w_func

    @property
    def tag_pairing_rule(self) -> TagPairingRule:
        return self._tpr

    # ---- plumbing -----------------------------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            h = C.c_void_p()
            N.check(N.lib().lchd_ctx_create(self._device, C.byref(h)))
            self._ctx = h
            if self._deterministic:
                N.check(N.lib().lchd_ctx_set_deterministic(h, 1))
        return self._ctx

    def _device_group(self):
        if self._group is None:
            h = C.c_void_p()
            devs = np.asarray(self._devices, dtype=np.int32)
            N.check(N.lib().lchd_group_create(N.ip(devs), len(devs), C.byref(h)))
            self._group = h
        return self._group

    def last_group_counts(self) -> List[int]:
        """Anchor pairs each device of ``devices`` scored in the most recent from_primitives / from_packed call."""
        if self._group is None:
            return []
        out = np.zeros(len(self._devices), dtype=np.int64)
        N.check(N.lib().lchd_group_last_counts(self._group, N.lp(out)))
        return out.tolist()

    def __del__(self):
        ctx, self._ctx = getattr(self, "_ctx", None), None
        grp, self._group = getattr(self, "_group", None), None
        try:
            if ctx is not None:
                N.lib().lchd_ctx_destroy(ctx)
            if grp is not None:
                N.lib().lchd_group_destroy(grp)
        except Exception:
            pass

    def _cats(self, seq) -> np.ndarray:
        if _fastpack is not None and hasattr(_fastpack, "cats_into"):  # the same loop in native code (Vec<String> extraction)
            out = np.empty(len(seq), dtype=np.int32)
            _fastpack.cats_into(seq, self._categories, out)
            return out
        get = self._categories.get
        return np.fromiter((get(str(s), -1) for s in seq), dtype=np.int32, count=len(seq))

    def _config(self, interner: Optional[Dict[str, int]] = None):
        """Build the lchd_config; returns (struct, keep-alive list)."""
        keep: List[Any] = []
        cfg = N.ConfigC()
        cfg.n_categories = len(self._weights)
        cfg.category_weights = N.dp(self._weights)
        wfs = list(self._w_func.values()) if isinstance(self._w_func, dict) else [self._w_func]
        arr = (N.WeightFunctionC * len(wfs))(*[w._c() for w in wfs])
        keep += [arr, wfs]
        cfg.n_weight_functions = len(wfs)
        cfg.weight_functions = arr
        cfg.sd_kind = self._sd._kind
        cfg.sd_n_params = len(self._sd._params)
        for i, v in enumerate(self._sd._params[:2]):
            cfg.sd_params[i] = v
        t = self._tpr
        cfg.tag_mode, cfg.tag_accept_same = t._mode, int(t._accept_same)
        cfg.tag_accepted_pairs, cfg.tag_ordered = int(t._accepted_pairs), int(t._ordered)
        interner = {} if interner is None else interner
        pairs = np.asarray([[interner.setdefault(a, len(interner)), interner.setdefault(b, len(interner))]
                            for a, b in sorted(t._pairs)], dtype=np.int32).reshape(-1, 2)
        keep.append(pairs)
        cfg.tag_pairs = N.ip(pairs) if len(pairs) else None
        cfg.n_tag_pairs = len(pairs)
        return cfg, keep

    def _wf_indices(self, keys: Optional[Sequence[str]], target_len: int) -> Optional[np.ndarray]:
        """keys_to_weight_functions, src/locohd.rs:230-283.  None => every pair uses weight function 0."""
        multiple = isinstance(self._w_func, dict)
        if multiple and keys is not None:
            if len(keys) != target_len:
                raise ValueError(f"The w_func_keys vector has an invalid length ({len(keys)} instead of {target_len})!")
            pos = {k: i for i, k in enumerate(self._wf_names)}
            bad = sum(1 for k in keys if k not in pos)
            if bad:
                raise ValueError(f"The vector contains {bad} out of {len(keys)} invalid weight function keys!")
            return np.fromiter((pos[k] for k in keys), dtype=np.int32, count=len(keys))
        if not multiple and keys is None:
            return None
        raise ValueError("Invalid pairing for the LoCoHD instance's w_func option and the method's w_func_keys parameter!")

    # ---- the four drivers -----------------------------------------------------------------------------
    def from_anchors(self, seq_a, seq_b, dists_a, dists_b, w_func_key: Optional[str] = None) -> float:
        """src/locohd.rs:392-406."""
        idx = self._wf_indices(None if w_func_key is None else [str(w_func_key)], 1)
        ca, cb = self._cats(list(seq_a)), self._cats(list(seq_b))
        da, db = _f64(dists_a).reshape(-1), _f64(dists_b).reshape(-1)
        cfg, keep = self._config()
        out = C.c_double()
        N.check(N.lib().lchd_from_anchors(self._context(), C.byref(cfg), N.ip(ca), ca.size, N.dp(da), da.size, N.ip(cb),
                                          cb.size, N.dp(db), db.size, 0 if idx is None else int(idx[0]),
                                          C.cast(C.byref(out), N._DP)))
        return out.value

    def from_dmxs(self, seq_a, seq_b, dmx_a, dmx_b, w_func_keys: Optional[Sequence[str]] = None) -> List[float]:
        """src/locohd.rs:410-458."""
        (ma, la), (mb, lb) = self._matrix(dmx_a), self._matrix(dmx_b)
        if ma.shape[0] != mb.shape[0]:  # :420-428
            raise ValueError(f"Expected matrices with the same length, got lengths {ma.shape[0]} and {mb.shape[0]}!")
        idx = self._wf_indices(None if w_func_keys is None else [str(k) for k in w_func_keys], ma.shape[0])
        ca, cb = self._cats(list(seq_a)), self._cats(list(seq_b))
        cfg, keep = self._config()
        out = np.empty(ma.shape[0])
        if ma.shape[0] == 0:
            return []
        if la is None and lb is None:
            N.check(N.lib().lchd_from_dmxs(self._context(), C.byref(cfg), N.ip(ca), ca.size, N.ip(cb), cb.size, N.dp(ma),
                                           ma.shape[0], ma.shape[1], N.dp(mb), mb.shape[0], mb.shape[1], N.ip(idx), N.dp(out)))
        else:  # ragged rows: every row with its own length (what lies beyond a row is never looked at, as in the reference)
            la = np.full(ma.shape[0], ma.shape[1], dtype=np.int32) if la is None else la
            lb = np.full(mb.shape[0], mb.shape[1], dtype=np.int32) if lb is None else lb
            N.check(N.lib().lchd_from_dmxs_ragged(self._context(), C.byref(cfg), N.ip(ca), ca.size, N.ip(cb), cb.size, N.dp(ma),
                                                  ma.shape[0], ma.shape[1], N.ip(la), N.dp(mb), mb.shape[0], mb.shape[1], N.ip(lb),
                                                  N.ip(idx), N.dp(out)))
        return out.tolist()

    def from_coords(self, seq_a, seq_b, coords_a, coords_b, w_func_keys: Optional[Sequence[str]] = None) -> List[float]:
        """src/locohd.rs:463-476."""
        xa, xb = self._coords(coords_a), self._coords(coords_b)
        if len(xa) != len(xb):
            raise ValueError(f"Expected matrices with the same length, got lengths {len(xa)} and {len(xb)}!")
        idx = self._wf_indices(None if w_func_keys is None else [str(k) for k in w_func_keys], len(xa))
        ca, cb = self._cats(list(seq_a)), self._cats(list(seq_b))
        cfg, keep = self._config()
        out = np.empty(len(xa))
        if len(xa) == 0:
            return []
        N.check(N.lib().lchd_from_coords(self._context(), C.byref(cfg), N.ip(ca), ca.size, N.ip(cb), cb.size, N.dp(xa), len(xa),
                                         N.dp(xb), len(xb), N.ip(idx), N.dp(out)))
        return out.tolist()

    def from_primitives(self, prim_a: Sequence[PrimitiveAtom], prim_b: Sequence[PrimitiveAtom], anchor_pairs,
                        threshold_distance: float) -> List[float]:
        """src/locohd.rs:479-567."""
        pairs, idx = self._anchor_arrays(anchor_pairs)
        pa, pb, interner = self._packed_lists(prim_a, prim_b)
        return self.from_packed(pa, pb, pairs, threshold_distance, wf_index=idx, interner=interner).tolist()

    # The reference's callers score the same structures over and over (one native structure against every decoy,
    # python_codes/casp14/casp14_extend_with_locohd.py:72-79; one reference frame against a trajectory,
    # python_codes/trajectory_analyzer.py:112-120), and extracting a few thousand Python objects costs several times the device
    # call.  The packed form of a list is therefore kept: an entry is valid while the list still holds the very same atom
    # objects (compared by identity in native code; the entry keeps them alive, so an address cannot be re-used) and no
    # PrimitiveAtom setter has run since.  Only plain lists / tuples of exactly PrimitiveAtom are cached.
    _CACHE_ENTRIES = 8

    def _cache_lookup(self, prims, parent):
        for k, e in enumerate(self._pack_cache):
            if e[0] is prims and e[3] is parent and e[2] == _ATOM_MUTATIONS[0] and _fastpack.same_items(prims, e[1]):
                if k:
                    self._pack_cache.insert(0, self._pack_cache.pop(k))
                return e
        return None

    def _type_map(self) -> np.ndarray:
        """primitive-type id (the process-wide table of PrimitiveAtom) -> category index of THIS instance, -1: not in its map"""
        m = self._pid_map
        if len(m) < len(_TYPE_NAMES):
            get = self._categories.get
            ext = [get(nm if type(nm) is str else str(nm), -1) for nm in _TYPE_NAMES[len(m):]]
            m = self._pid_map = np.concatenate([m, np.asarray(ext, dtype=np.int32)])
        return m

    def _pack_global(self, prims) -> _Packed:
        """Packed form with tags from the process-wide table: the gather over interned ids when the list holds exactly PrimitiveAtom
        objects, else the general extraction (strings looked up per atom) into the same id space."""
        n = len(prims)
        if _fastpack is not None and hasattr(_fastpack, "pack_atoms"):
            xyz, cat, tag = np.empty((n, 3), dtype=np.float64), np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
            if _fastpack.pack_atoms(prims, PrimitiveAtom, self._type_map(), xyz, cat, tag):
                return _Packed(xyz, cat, tag)
        return self.pack(prims, _TAG_IDS)

    def _packed_lists(self, prim_a, prim_b):
        """(packed A, packed B, interner): tags of both lists (and of the tag rule, _config) are ids of ONE table, the process-wide
        one of PrimitiveAtom -- only their equality matters (tag_pairing_rule.rs:49-75)."""
        if _fastpack is None or not hasattr(_fastpack, "same_items"):
            return self.pack(prim_a, _TAG_IDS), self.pack(prim_b, _TAG_IDS), _TAG_IDS
        entries = []
        for prims in (prim_a, prim_b):
            e = self._cache_lookup(prims, None)
            if e is None:
                packed = self._pack_global(prims)
                items = _fastpack.items_tuple(prims, PrimitiveAtom)
                e = (prims, items, _ATOM_MUTATIONS[0], None, packed, _TAG_IDS)
                if items is not None:
                    self._pack_cache.insert(0, e)
                    del self._pack_cache[self._CACHE_ENTRIES:]
            entries.append(e)
        return entries[0][4], entries[1][4], _TAG_IDS

    def _anchor_arrays(self, anchor_pairs):
        """([P][2] int64 anchors, weight-function indices or None); the conversion of an unchanged list of tuples is kept."""
        c = self._anchor_cache
        if c is not None and c[0] is anchor_pairs and _fastpack is not None and hasattr(_fastpack, "same_items") and _fastpack.same_items(anchor_pairs, c[1]):
            return c[2], c[3]
        if _fastpack is not None and hasattr(_fastpack, "pairs_into") and hasattr(anchor_pairs, "__len__"):
            arr = np.empty((len(anchor_pairs), 2), dtype=np.int64)
            keys = _fastpack.pairs_into(anchor_pairs, arr)  # (AnchorPairSpecifier in native code, like the PyO3 derive)
            idx = self._wf_indices(keys, len(arr))
        else:
            pairs, keys = self._split_anchor_pairs(anchor_pairs)
            idx = self._wf_indices(keys, len(pairs))
            arr = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
        items = _fastpack.items_tuple(anchor_pairs, tuple) if (_fastpack is not None and hasattr(_fastpack, "items_tuple")) else None
        self._anchor_cache = (anchor_pairs, items, arr, idx) if items is not None else None
        return arr, idx

    def from_primitives_batch(self, structures: Sequence[Sequence[PrimitiveAtom]],
                              jobs: Sequence[Tuple[int, int, Sequence[Tuple[int, int]]]],
                              threshold_distance: float) -> List[List[float]]:
        """Additive: many `from_primitives` calls in one device pass.  `structures` are lists of
        PrimitiveAtoms; a job `(a, b, anchor_pairs)` asks for `from_primitives(structures[a], structures[b],
        anchor_pairs, threshold_distance)`.  Returns one score list per job, each bit-identical to the single call.
        Replaces loops such as python_codes/casp14/casp14_extend_with_locohd.py:42-88 (every decoy against the native
        structure) or python_codes/ensembles/compare_ensembles.py:277-296: all structures are uploaded once as one batch
        object, the anchor pairs of all jobs are scored by one kernel sequence."""
        from .device import DeviceSession  # torch is only needed on this path

        if isinstance(self._w_func, dict):
            raise ValueError("from_primitives_batch needs a single weight function (no per-pair keys)")
        interner: Dict[str, int] = {}
        packed = [self.pack(st, interner) for st in structures]
        sizes = [len(pk.cat) for pk in packed]
        flat, spans = [], []
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        for a, b, pairs in jobs:
            if not (0 <= int(a) < len(packed) and 0 <= int(b) < len(packed)):
                raise IndexError(f"job refers to structure {a} / {b} of {len(packed)}")
            pr = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
            if len(pr) and (pr.min() < 0 or pr[:, 0].max() >= sizes[int(a)] or pr[:, 1].max() >= sizes[int(b)]):
                raise N.PanicException("index out of bounds: an anchor index is outside its structure (src/locohd.rs:521)")
            spans.append((len(flat), len(pr)))
            flat.extend([pr + np.asarray([offsets[int(a)], offsets[int(b)]], dtype=np.int64)] if len(pr) else [])
        total = sum(n for _, n in spans)
        if total == 0:
            return [[] for _ in jobs]
        sess = DeviceSession(self, device=None if self._device < 0 else self._device, interner=interner)
        try:
            torch = sess.torch
            batch, _ = sess.upload_batch([(pk.xyz, pk.cat, pk.tag) for pk in packed])
            anchors = torch.from_numpy(np.concatenate(flat)).to(torch.device("cuda", sess.device))
            scores = sess.from_primitives(batch, batch, anchors, float(threshold_distance)).cpu().numpy()
        finally:
            sess.close()
        out, pos = [], 0
        for _, n in spans:
            out.append(scores[pos:pos + n].tolist())
            pos += n
        return out

    # ---- additive array-level entry (no per-atom Python objects; used by bench.py and batch callers) --------
    def pack(self, prims: Sequence[PrimitiveAtom], interner: Optional[Dict[str, int]] = None) -> _Packed:
        interner = {} if interner is None else interner
        n = len(prims)
        if _fastpack is not None:  # native extraction (what PyO3 does for the reference), same results as the loop below
            xyz, cat, tag = np.empty((n, 3), dtype=np.float64), np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
            _fastpack.pack_into(prims, self._categories, interner, xyz, cat, tag)
            return _Packed(xyz, cat, tag)
        xyz = np.array([p._coordinates if type(p) is PrimitiveAtom else p.coordinates for p in prims], dtype=np.float64).reshape(n, 3)
        cat = self._cats([p.primitive_type for p in prims])
        intern = interner.setdefault
        tag = np.fromiter((intern(p.tag, len(interner)) for p in prims), dtype=np.int32, count=n)
        return _Packed(xyz, cat, tag)

    def from_packed(self, pa: _Packed, pb: _Packed, pairs, threshold_distance: float, wf_index: Optional[np.ndarray] = None,
                    interner: Optional[Dict[str, int]] = None) -> np.ndarray:
        anchors = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
        cfg, keep = self._config(interner)
        out = np.empty(len(anchors))
        if len(anchors) == 0:
            return out
        xa, xb = _f64(pa.xyz).reshape(-1, 3), _f64(pb.xyz).reshape(-1, 3)
        if self._devices is not None and len(self._devices) > 1:  # the instance's device group (the reference: its thread pool)
            N.check(N.lib().lchd_group_from_primitives(self._device_group(), C.byref(cfg), N.dp(xa), N.ip(pa.cat), N.ip(pa.tag), len(xa),
                                                       N.dp(xb), N.ip(pb.cat), N.ip(pb.tag), len(xb), N.lp(anchors), N.ip(wf_index),
                                                       len(anchors), float(threshold_distance), N.dp(out)))
            return out
        N.check(N.lib().lchd_from_primitives(self._context(), C.byref(cfg), N.dp(xa), N.ip(pa.cat), N.ip(pa.tag), len(xa),
                                             N.dp(xb), N.ip(pb.cat), N.ip(pb.tag), len(xb), N.lp(anchors), N.ip(wf_index),
                                             len(anchors), float(threshold_distance), N.dp(out)))
        return out

    # ---- argument conversion helpers ------------------------------------------------------------------------
    @staticmethod
    def _matrix(m):
        """Vec<Vec<f64>> -> (rectangular f64 matrix, row lengths or None).  Ragged rows (the reference sorts each row with a
        prefix of seq, utils.rs:25-39) are padded and travel with their lengths (lchd_from_dmxs_ragged)."""
        lens = None
        try:
            arr = _f64(m)
        except ValueError:
            rows = [_f64(r).reshape(-1) for r in m]
            width = max((len(r) for r in rows), default=0)
            if any(len(r) == 0 for r in rows):
                raise N.PanicException("index out of bounds: empty distance row (src/locohd.rs:74)")
            arr = np.zeros((len(rows), width))
            for k, r in enumerate(rows):
                arr[k, : len(r)] = r
            lens = np.asarray([len(r) for r in rows], dtype=np.int32)
        if arr.ndim == 1 and arr.size == 0:
            arr = arr.reshape(0, 0)
        if arr.ndim != 2:
            raise TypeError("a distance matrix must be a 2-D sequence of floats")
        return arr, lens

    @staticmethod
    def _coords(x) -> np.ndarray:
        arr = _f64(x)
        if arr.size == 0:
            return arr.reshape(0, 3)
        if arr.ndim != 2 or arr.shape[1] != 3:
            raise TypeError("coordinates must be a sequence of [x, y, z] triples")
        return arr

    @staticmethod
    def _split_anchor_pairs(anchor_pairs) -> Tuple[List[Tuple[int, int]], Optional[List[str]]]:
        """AnchorPairSpecifier (src/locohd.rs:34-40): 3-tuples are tried first, so an EMPTY list is the
        with-key variant (and then fails against a single weight function, :276-281)."""
        ap = list(anchor_pairs)
        if len(ap) == 0 or all(len(p) == 3 for p in ap):
            pairs, keys = [(int(p[0]), int(p[1])) for p in ap], [str(p[2]) for p in ap]
        elif all(len(p) == 2 for p in ap):
            pairs, keys = [(int(p[0]), int(p[1])) for p in ap], None
        else:
            raise TypeError("failed to extract enum AnchorPairSpecifier ('WithWeightFunctionKey | WithoutWeightFunctionKey')")
        if any(a < 0 or b < 0 for a, b in pairs):
            raise OverflowError("can't convert negative int to unsigned")
        return pairs, keys
