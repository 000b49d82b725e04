"""Device-resident scoring session (additive API; what bench.py, the multi-GPU sharding and batch callers use).

The reference API is one structure pair per call with everything copied in and out
(/root/reference/src/locohd.rs:479-567).  A `DeviceSession` keeps the two structures in HBM as SoA arrays,
takes anchor pairs / scores as torch CUDA tensors (torch is only the allocator and the stream provider
here) and runs the same kernels through `lchd_from_primitives_dev`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _native as N
from .api import LoCoHD


class DeviceSession:
    def __init__(self, lchd: LoCoHD, device: Optional[int] = None, interner: Optional[dict] = None):
        import torch

        self.torch = torch
        if not torch.cuda.is_available():
            raise N.DeviceError("no usable HIP device: loco_hd_amd has no CPU fallback for the scoring path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        torch.cuda.set_device(self.device)
        self.lchd = lchd
        self._ctx = C.c_void_p()
        N.check(N.lib().lchd_ctx_create(self.device, C.byref(self._ctx)))
        self._cfg, self._keep = lchd._config(interner)
        N.check(N.lib().lchd_ctx_set_config(self._ctx, C.byref(self._cfg)))
        self.use_current_stream()
        self._clouds = []

    def use_current_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        N.check(N.lib().lchd_ctx_set_stream(self._ctx, C.c_void_p(s)))

    def enable_timing(self, on: bool = True):
        N.lib().lchd_ctx_enable_timing(self._ctx, int(on))

    def last_ms(self) -> dict:
        return {k: N.lib().lchd_ctx_last_ms(self._ctx, k.encode()) for k in ("cells", "anchors", "env", "sweep")}

    def last_env_points(self) -> int:
        return int(N.lib().lchd_ctx_last_env_points(self._ctx))

    def upload(self, xyz: np.ndarray, cat: np.ndarray, tag: Optional[np.ndarray] = None):
        """Put one structure (xyz [n][3] f64, category ids, interned tags) into HBM; returns an opaque handle."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        cat = np.ascontiguousarray(cat, dtype=np.int32)
        tag = np.zeros(len(cat), dtype=np.int32) if tag is None else np.ascontiguousarray(tag, dtype=np.int32)
        h = C.c_void_p()
        N.check(N.lib().lchd_cloud_create(self._ctx, N.dp(xyz), N.ip(cat), N.ip(tag), len(xyz), C.byref(h)))
        self._clouds.append(h)
        return h

    def upload_batch(self, structures):
        """Put MANY structures into HBM as one object.  `structures` = sequence of (xyz [n_k][3], cat [n_k], tag [n_k] or
        None).  Returns (handle, offsets): atom j of structure k is global atom offsets[k] + j -- the index to use in the
        anchor tensor; environments never mix atoms of different structures."""
        xs, cs, ts, ss, offs = [], [], [], [], [0]
        for k, st in enumerate(structures):
            xyz = np.ascontiguousarray(st[0], dtype=np.float64).reshape(-1, 3)
            cat = np.ascontiguousarray(st[1], dtype=np.int32)
            tag = np.zeros(len(cat), dtype=np.int32) if len(st) < 3 or st[2] is None else np.ascontiguousarray(st[2], dtype=np.int32)
            xs.append(xyz); cs.append(cat); ts.append(tag); ss.append(np.full(len(cat), k, dtype=np.int32))
            offs.append(offs[-1] + len(cat))
        xyz, cat, tag, sid = np.concatenate(xs), np.concatenate(cs), np.concatenate(ts), np.concatenate(ss)
        h = C.c_void_p()
        N.check(N.lib().lchd_cloud_create_batch(self._ctx, N.dp(xyz), N.ip(cat), N.ip(tag), N.ip(sid), len(xyz), len(structures),
                                                C.byref(h)))
        self._clouds.append(h)
        return h, np.asarray(offs, dtype=np.int64)

    def set_coords(self, cloud, xyz: np.ndarray):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        N.check(N.lib().lchd_cloud_set_coords(self._ctx, cloud, N.dp(xyz)))

    def from_primitives(self, cloud_a, cloud_b, anchors, threshold_distance: float, out=None, wf_index=None):
        """anchors: torch int64 CUDA tensor [P][2]; returns (or fills) a torch float64 CUDA tensor [P]."""
        torch = self.torch
        assert anchors.is_cuda and anchors.dtype == torch.int64 and anchors.is_contiguous()
        p = anchors.shape[0]
        if out is None:
            out = torch.empty(p, dtype=torch.float64, device=anchors.device)
        assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() >= p
        wf_ptr = None
        if wf_index is not None:
            assert wf_index.is_cuda and wf_index.dtype == torch.int32 and wf_index.is_contiguous()
            wf_ptr = C.c_void_p(wf_index.data_ptr())
        N.check(N.lib().lchd_from_primitives_dev(self._ctx, cloud_a, cloud_b, C.c_void_p(anchors.data_ptr()), wf_ptr, p,
                                                 float(threshold_distance), C.c_void_p(out.data_ptr())))
        return out

    def close(self):
        if self._ctx:
            for h in self._clouds:
                N.lib().lchd_cloud_destroy(self._ctx, h)
            self._clouds = []
            N.lib().lchd_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
