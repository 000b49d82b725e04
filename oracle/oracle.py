"""ctypes front-end of the CPU oracle (oracle/locohd_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this module; the
product package ``loco_hd_amd`` never does.  The classes mirror the reference's Python surface
(/root/reference/loco_hd/loco_hd.pyi) closely enough that the parity tests read like the reference's
own tests; strings are interned here exactly as the reference's HashMap would index them
(src/locohd.rs:312-316).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
# LCHD_ASAN=1: the AddressSanitizer / UBSan build of the checker (make -C oracle asan; README "Sanitizer run")
_ASAN = bool(os.environ.get("LCHD_ASAN"))
_SO = _DIR / ("asan/liblocohd_oracle.so" if _ASAN else "liblocohd_oracle.so")

WF_KINDS = {"hyper_exp": 0, "dagum": 1, "uniform": 2, "kumaraswamy": 3}
SD_KINDS = {"Hellinger": 0, "Kolmogorov-Smirnov": 1, "Kullback-Leibler": 2, "Renyi": 3}


class OraclePanic(RuntimeError):
    """Stands in for pyo3_runtime.PanicException (a Rust panic in the reference)."""


def build(force: bool = False) -> Path:
    src = _DIR / "locohd_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_DIR), "-B", "asan" if _ASAN else "liblocohd_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class _Cfg(C.Structure):
    _fields_ = [
        ("n_categories", C.c_int32),
        ("category_weights", C.POINTER(C.c_double)),
        ("sd_kind", C.c_int32),
        ("sd_params", C.c_double * 2),
        ("tag_mode", C.c_int32),
        ("tag_accept_same", C.c_int32),
        ("tag_accepted_pairs", C.c_int32),
        ("tag_ordered", C.c_int32),
        ("tag_pairs", C.POINTER(C.c_int32)),
        ("n_tag_pairs", C.c_int64),
    ]


class _Wf(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_params", C.c_int32), ("params", C.POINTER(C.c_double))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_SO))
        _lib.orc_last_error.restype = C.c_char_p
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _check(rc):
    if rc == 0:
        return
    msg = lib().orc_last_error().decode()
    if rc == 1:
        raise ValueError(msg)
    raise OraclePanic(msg)


class WeightFunction:
    def __init__(self, function_name, parameters):
        self.function_name = str(function_name)
        self.parameters = [float(x) for x in parameters]
        if self.function_name not in WF_KINDS:
            raise ValueError(f'No function implemented with name "{self.function_name}"!')
        self._kind = WF_KINDS[self.function_name]
        self._p = np.asarray(self.parameters, dtype=np.float64)
        _check(lib().orc_wf_validate(self._kind, _dp(self._p), len(self._p)))

    def integral_point(self, point):
        out = C.c_double()
        _check(lib().orc_wf_integral_point(self._kind, _dp(self._p), len(self._p), C.c_double(point), C.byref(out)))
        return out.value

    def integral_vec(self, points):
        return [self.integral_point(float(x)) for x in points]

    def integral_range(self, point_from, point_to):
        out = C.c_double()
        _check(lib().orc_wf_integral_range(self._kind, _dp(self._p), len(self._p), C.c_double(point_from),
                                           C.c_double(point_to), C.byref(out)))
        return out.value

    def _c(self):
        return _Wf(self._kind, len(self._p), _dp(self._p))


class StatisticalDistance:
    def __init__(self, distance_name, parameters):
        if distance_name not in SD_KINDS:
            raise ValueError(f"Invalid statistical distance name {distance_name}!")
        self._kind = SD_KINDS[distance_name]
        self._p = np.asarray(list(parameters), dtype=np.float64)
        _check(lib().orc_sd_validate(self._kind, len(self._p)))

    def run(self, p1, p2):
        p1 = np.ascontiguousarray(p1, dtype=np.float64)
        p2 = np.ascontiguousarray(p2, dtype=np.float64)
        n = min(len(p1), len(p2))  # zip() semantics
        prm = np.zeros(2)
        prm[: len(self._p)] = self._p
        out = C.c_double()
        _check(lib().orc_sd_run(self._kind, _dp(prm), _dp(p1), _dp(p2), n, C.byref(out)))
        return out.value


class TagPairingRule:
    def __init__(self, variant):
        if "accept_same" in variant:
            self.mode, self.accept_same = 0, bool(variant["accept_same"])
            self.tag_pairs, self.accepted_pairs, self.ordered = set(), True, True
        else:
            self.mode, self.accept_same = 1, True
            self.tag_pairs = {(str(a), str(b)) for a, b in variant["tag_pairs"]}
            self.accepted_pairs, self.ordered = bool(variant["accepted_pairs"]), bool(variant["ordered"])

    def pair_accepted(self, pair):
        interner = {}
        cfg, keep = self._cfg_fields(interner)
        t0 = interner.setdefault(pair[0], len(interner))
        t1 = interner.setdefault(pair[1], len(interner))
        c = _Cfg()
        for k, v in cfg.items():
            setattr(c, k, v)
        return bool(lib().orc_tag_pair_accepted(C.byref(c), t0, t1))

    def _cfg_fields(self, interner):
        pairs = np.asarray(
            [[interner.setdefault(a, len(interner)), interner.setdefault(b, len(interner))] for a, b in sorted(self.tag_pairs)],
            dtype=np.int32,
        ).reshape(-1, 2)
        return dict(tag_mode=self.mode, tag_accept_same=int(self.accept_same), tag_accepted_pairs=int(self.accepted_pairs),
                    tag_ordered=int(self.ordered), tag_pairs=_ip(pairs), n_tag_pairs=len(pairs)), pairs


class PrimitiveAtom:
    def __init__(self, primitive_type, tag, coordinates):
        self.primitive_type, self.tag = primitive_type, tag
        self.coordinates = [float(x) for x in coordinates]


class LoCoHD:
    """Oracle twin of loco_hd.LoCoHD (src/locohd.rs:286-568)."""

    def __init__(self, categories, w_func=None, tag_pairing_rule=None, n_of_threads=None, category_weights=None,
                 statistical_distance=None):
        categories = [str(c) for c in categories]
        self.categories = {}
        for i, name in enumerate(categories):  # HashMap collect: later duplicates overwrite (:312-316)
            self.categories[name] = i
        n_map = len(self.categories)
        w = np.ones(n_map) if category_weights is None else np.asarray(list(category_weights), dtype=np.float64)
        _check(lib().orc_config_validate(len(categories), n_map, _dp(w), len(w)))
        self.category_weights = w
        # NB: with duplicate names the surviving indices may exceed n_map-1 in the reference (index = position
        # in the original list); the oracle keeps that quirk out of scope and requires unique names.
        self.w_func = WeightFunction("uniform", [3.0, 10.0]) if w_func is None else w_func
        self.tag_pairing_rule = TagPairingRule({"accept_same": True}) if tag_pairing_rule is None else tag_pairing_rule
        self.statistical_distance = StatisticalDistance("Hellinger", [2.0]) if statistical_distance is None else statistical_distance
        self.n_threads = int(n_of_threads) if n_of_threads else (os.cpu_count() or 1)

    # -- helpers ----------------------------------------------------------------------------------
    def _cats(self, seq):
        return np.asarray([self.categories.get(str(s), -1) for s in seq], dtype=np.int32)

    def _cfg(self, interner=None):
        c = _Cfg()
        c.n_categories = len(self.category_weights)
        c.category_weights = _dp(self.category_weights)
        c.sd_kind = self.statistical_distance._kind
        for i, v in enumerate(self.statistical_distance._p):
            c.sd_params[i] = v
        fields, keep = self.tag_pairing_rule._cfg_fields({} if interner is None else interner)
        for k, v in fields.items():
            setattr(c, k, v)
        return c, keep

    def _wfs(self, keys, target_len):
        """keys_to_weight_functions, src/locohd.rs:230-283"""
        if isinstance(self.w_func, dict) and keys is not None:
            if len(keys) != target_len:
                raise ValueError(f"The w_func_keys vector has an invalid length ({len(keys)} instead of {target_len})!")
            names = list(self.w_func)
            bad = sum(1 for k in keys if k not in self.w_func)
            if bad:
                raise ValueError(f"The vector contains {bad} out of {len(keys)} invalid weight function keys!")
            arr = (_Wf * len(names))(*[self.w_func[n]._c() for n in names])
            idx = np.asarray([names.index(k) for k in keys], dtype=np.int32)
            return arr, idx
        if not isinstance(self.w_func, dict) and keys is None:
            return (_Wf * 1)(self.w_func._c()), None
        raise ValueError("Invalid pairing for the LoCoHD instance's w_func option and the method's w_func_keys parameter!")

    # -- the four drivers -------------------------------------------------------------------------
    def from_anchors(self, seq_a, seq_b, dists_a, dists_b, w_func_key=None):
        wfs, idx = self._wfs(None if w_func_key is None else [w_func_key], 1)
        ca, cb = self._cats(seq_a), self._cats(seq_b)
        da, db = np.ascontiguousarray(dists_a, dtype=np.float64), np.ascontiguousarray(dists_b, dtype=np.float64)
        cfg, keep = self._cfg()
        out = C.c_double()
        wf = wfs[0 if idx is None else int(idx[0])]
        _check(lib().orc_stat_dist_integral(C.byref(cfg), _ip(ca), _dp(da), C.c_int64(len(ca)), C.c_int64(len(da)), _ip(cb),
                                            _dp(db), C.c_int64(len(cb)), C.c_int64(len(db)), C.byref(wf), C.byref(out)))
        return out.value

    def from_dmxs(self, seq_a, seq_b, dmx_a, dmx_b, w_func_keys=None):
        ma, mb = np.ascontiguousarray(dmx_a, dtype=np.float64), np.ascontiguousarray(dmx_b, dtype=np.float64)
        ma, mb = ma.reshape(len(ma), -1), mb.reshape(len(mb), -1)
        if len(ma) != len(mb):
            raise ValueError(f"Expected matrices with the same length, got lengths {len(ma)} and {len(mb)}!")
        wfs, idx = self._wfs(w_func_keys, len(ma))
        ca, cb = self._cats(seq_a), self._cats(seq_b)
        cfg, keep = self._cfg()
        out = np.zeros(len(ma))
        _check(lib().orc_from_dmxs(C.byref(cfg), _ip(ca), C.c_int64(len(ca)), _ip(cb), C.c_int64(len(cb)), _dp(ma),
                                   C.c_int64(ma.shape[0]), C.c_int64(ma.shape[1]), _dp(mb), C.c_int64(mb.shape[0]),
                                   C.c_int64(mb.shape[1]), wfs, None if idx is None else _ip(idx), self.n_threads, _dp(out)))
        return out.tolist()

    def from_coords(self, seq_a, seq_b, coords_a, coords_b, w_func_keys=None):
        xa = np.ascontiguousarray(coords_a, dtype=np.float64).reshape(-1, 3)
        xb = np.ascontiguousarray(coords_b, dtype=np.float64).reshape(-1, 3)
        if len(xa) != len(xb):
            raise ValueError(f"Expected matrices with the same length, got lengths {len(xa)} and {len(xb)}!")
        wfs, idx = self._wfs(w_func_keys, len(xa))
        ca, cb = self._cats(seq_a), self._cats(seq_b)
        cfg, keep = self._cfg()
        out = np.zeros(len(xa))
        _check(lib().orc_from_coords(C.byref(cfg), _ip(ca), C.c_int64(len(ca)), _ip(cb), C.c_int64(len(cb)), _dp(xa),
                                     C.c_int64(len(xa)), _dp(xb), C.c_int64(len(xb)), wfs, None if idx is None else _ip(idx),
                                     self.n_threads, _dp(out)))
        return out.tolist()

    def from_primitives(self, prim_a, prim_b, anchor_pairs, threshold_distance, return_env_sizes=False):
        anchor_pairs = list(anchor_pairs)
        if len(anchor_pairs) == 0 or len(anchor_pairs[0]) == 3:  # an empty list matches the 3-tuple variant (:34-40)
            keys = [p[2] for p in anchor_pairs]
            pairs = [(p[0], p[1]) for p in anchor_pairs]
        else:
            keys, pairs = None, anchor_pairs
        wfs, idx = self._wfs(keys, len(pairs))
        return self.from_arrays(*self.pack(prim_a), *self.pack(prim_b), pairs, threshold_distance, wfs=wfs, wf_idx=idx,
                                return_env_sizes=return_env_sizes)

    def pack(self, prims, interner=None):
        self._interner = getattr(self, "_interner", {}) if interner is None else interner
        xyz = np.asarray([p.coordinates for p in prims], dtype=np.float64).reshape(-1, 3)
        cat = self._cats([p.primitive_type for p in prims])
        tag = np.asarray([self._interner.setdefault(p.tag, len(self._interner)) for p in prims], dtype=np.int32)
        return xyz, cat, tag

    def from_arrays(self, xyz_a, cat_a, tag_a, xyz_b, cat_b, tag_b, pairs, threshold_distance, wfs=None, wf_idx=None,
                    return_env_sizes=False, interner=None):
        """SoA entry used by the bench/parity harness: integer categories and tags, [P][2] anchors."""
        if wfs is None:
            wfs, wf_idx = self._wfs(None, len(pairs))
        cfg, keep = self._cfg(getattr(self, "_interner", None) if interner is None else interner)
        xyz_a, xyz_b = np.ascontiguousarray(xyz_a, dtype=np.float64), np.ascontiguousarray(xyz_b, dtype=np.float64)
        cat_a, cat_b = np.ascontiguousarray(cat_a, dtype=np.int32), np.ascontiguousarray(cat_b, dtype=np.int32)
        tag_a, tag_b = np.ascontiguousarray(tag_a, dtype=np.int32), np.ascontiguousarray(tag_b, dtype=np.int32)
        anchors = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
        out = np.zeros(len(anchors))
        sizes = np.zeros((len(anchors), 2), dtype=np.int64)
        _check(lib().orc_from_primitives(C.byref(cfg), _dp(xyz_a), _ip(cat_a), _ip(tag_a), C.c_int64(len(xyz_a)), _dp(xyz_b),
                                         _ip(cat_b), _ip(tag_b), C.c_int64(len(xyz_b)), _lp(anchors), C.c_int64(len(anchors)),
                                         wfs, None if wf_idx is None else _ip(wf_idx), C.c_double(threshold_distance),
                                         self.n_threads, _dp(out), _lp(sizes)))
        if return_env_sizes:
            return out, sizes
        return out.tolist()
