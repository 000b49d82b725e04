"""ctypes binding of libloco_hd_hip.so (the C ABI in include/loco_hd_hip.h).

This is the stub a maintainer of the reference would write instead of the PyO3 module
`loco_hd.loco_hd` (/root/reference/src/lib.rs:9-17).  The library is built in-tree by
`make -C loco_hd_amd/csrc` (or `__graft_entry__.build()`); if it is missing, importing any scoring
entry point fails loudly -- there is no Python/CPU fallback for the scoring path.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["LCHD_LIB"]) if os.environ.get("LCHD_LIB") else _PKG / "libloco_hd_hip.so"  # LCHD_LIB: tuning builds

OK, EVALUE, EPANIC, EDEVICE, EUNSUPPORTED = 0, 1, 2, 3, 4


class PanicException(RuntimeError):
    """Raised where the reference's Rust core would panic (pyo3_runtime.PanicException)."""


class DeviceError(RuntimeError):
    """HIP failure or no usable MI355X: the scoring path has no CPU fallback."""


class WeightFunctionC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_params", C.c_int32), ("params", C.POINTER(C.c_double))]


class ConfigC(C.Structure):
    _fields_ = [
        ("n_categories", C.c_int32),
        ("category_weights", C.POINTER(C.c_double)),
        ("n_weight_functions", C.c_int32),
        ("weight_functions", C.POINTER(WeightFunctionC)),
        ("sd_kind", C.c_int32),
        ("sd_n_params", C.c_int32),
        ("sd_params", C.c_double * 2),
        ("tag_mode", C.c_int32),
        ("tag_accept_same", C.c_int32),
        ("tag_accepted_pairs", C.c_int32),
        ("tag_ordered", C.c_int32),
        ("tag_pairs", C.POINTER(C.c_int32)),
        ("n_tag_pairs", C.c_int64),
    ]


_DP, _IP, _LP, _VP = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double

_PROTOS = {
    "lchd_last_error": (C.c_char_p, []),
    "lchd_version": (C.c_char_p, []),
    "lchd_wf_validate": (C.c_int, [_i32, _DP, _i32]),
    "lchd_wf_cdf": (C.c_int, [_i32, _DP, _i32, _DP, _i64, _DP]),
    "lchd_sd_validate": (C.c_int, [_i32, _i32]),
    "lchd_sd_run": (C.c_int, [_i32, _DP, _DP, _DP, _i32, _DP]),
    "lchd_config_validate": (C.c_int, [_i64, _i64, _DP, _i64]),
    "lchd_ctx_create": (C.c_int, [_i32, C.POINTER(_VP)]),
    "lchd_ctx_destroy": (None, [_VP]),
    "lchd_ctx_set_stream": (C.c_int, [_VP, _VP]),
    "lchd_ctx_set_config": (C.c_int, [_VP, C.POINTER(ConfigC)]),
    "lchd_from_anchors": (C.c_int, [_VP, C.POINTER(ConfigC), _IP, _i64, _DP, _i64, _IP, _i64, _DP, _i64, _i32, _DP]),
    "lchd_from_dmxs": (C.c_int, [_VP, C.POINTER(ConfigC), _IP, _i64, _IP, _i64, _DP, _i64, _i64, _DP, _i64, _i64, _IP, _DP]),
    "lchd_from_dmxs_ragged": (C.c_int, [_VP, C.POINTER(ConfigC), _IP, _i64, _IP, _i64, _DP, _i64, _i64, _IP, _DP, _i64, _i64, _IP, _IP, _DP]),
    "lchd_from_coords": (C.c_int, [_VP, C.POINTER(ConfigC), _IP, _i64, _IP, _i64, _DP, _i64, _DP, _i64, _IP, _DP]),
    "lchd_from_primitives": (C.c_int, [_VP, C.POINTER(ConfigC), _DP, _IP, _IP, _i64, _DP, _IP, _IP, _i64, _LP, _IP, _i64, _f64, _DP]),
    "lchd_cloud_create": (C.c_int, [_VP, _DP, _IP, _IP, _i64, C.POINTER(_VP)]),
    "lchd_cloud_create_batch": (C.c_int, [_VP, _DP, _IP, _IP, _IP, _i64, _i32, C.POINTER(_VP)]),
    "lchd_cloud_size": (C.c_int64, [_VP]),
    "lchd_cloud_set_coords": (C.c_int, [_VP, _VP, _DP]),
    "lchd_cloud_destroy": (None, [_VP, _VP]),
    "lchd_from_primitives_dev": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _i64, _f64, _VP]),
    "lchd_from_primitives_dev_async": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _i64, _f64, _VP]),
    "lchd_ctx_finish": (C.c_int, [_VP]),
    "lchd_from_coords_dev": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "lchd_frames_create": (C.c_int, [_VP, _VP, _i32, C.POINTER(_VP)]),
    "lchd_frames_load": (C.c_int, [_VP, _VP, _DP, _i32, _VP]),
    "lchd_frames_set_sources": (C.c_int, [_VP, _VP, _IP, _IP, _i64]),
    "lchd_frames_load_atoms": (C.c_int, [_VP, _VP, C.POINTER(C.c_float), _i32, _VP]),
    "lchd_frames_load_atoms_dev": (C.c_int, [_VP, _VP, _VP, _i32, _VP]),
    "lchd_frames_last_convert_ms": (C.c_double, [_VP, _VP]),
    "lchd_cloud_get_coords": (C.c_int, [_VP, _VP, _DP, _i64]),
    "lchd_group_create": (C.c_int, [_IP, _i32, C.POINTER(_VP)]),
    "lchd_group_destroy": (None, [_VP]),
    "lchd_group_size": (_i32, [_VP]),
    "lchd_group_from_primitives": (C.c_int, [_VP, C.POINTER(ConfigC), _DP, _IP, _IP, _i64, _DP, _IP, _IP, _i64, _LP, _IP, _i64, _f64, _DP]),
    "lchd_group_last_counts": (C.c_int, [_VP, _LP]),
    "lchd_shard_plan_dev": (C.c_int, [_VP, _VP, _i64, _i64, _i64, _i32, _LP]),
    "lchd_shard_select_dev": (C.c_int, [_VP, _VP, _i64, _i64, _i64, _i32, _VP, _VP]),
    "lchd_unshard_scores_dev": (C.c_int, [_VP, _VP, _LP, _i32, _i64, _VP, _i64]),
    "lchd_ctx_enable_timing": (C.c_int, [_VP, _i32]),
    "lchd_ctx_last_ms": (C.c_double, [_VP, C.c_char_p]),
    "lchd_ctx_last_env_points": (C.c_int64, [_VP]),
    "lchd_ctx_last_dense_fused": (C.c_int32, [_VP]),
    "lchd_ctx_set_deterministic": (C.c_int, [_VP, _i32]),
    "lchd_ctx_get_deterministic": (C.c_int32, [_VP]),
    "lchd_ctx_pass_count": (C.c_int64, [_VP]),
    "lchd_ctx_subset_pass_count": (C.c_int64, [_VP]),
    "lchd_ctx_per_pair_pass_count": (C.c_int64, [_VP]),
    "lchd_ctx_last_store_bytes": (C.c_int64, [_VP]),
}

_lib = None


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so with the same soname as /opt/rocm's.  A process can only
    hold one of them: if ours pulled in the system runtime first, a later `import torch` would find no GPU.  So when
    torch is installed but not imported yet, map its bundled runtime first; our library then binds to it (exactly what
    happens when torch is imported first)."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the HIP core.  Raises ImportError if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {_PKG / 'csrc'}` (hipcc --offload-arch=gfx950). "
                "loco_hd_amd has no CPU fallback for the scoring path."
            )
        _preload_torch_hip_runtime()
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc == OK:
        return
    msg = lib().lchd_last_error().decode(errors="replace")
    if rc == EVALUE:
        raise ValueError(msg)
    if rc == EPANIC:
        raise PanicException(msg)
    if rc == EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise DeviceError(msg)


def dp(a):
    return a.ctypes.data_as(_DP)


def ip(a):
    return None if a is None else a.ctypes.data_as(_IP)


def lp(a):
    return a.ctypes.data_as(_LP)
