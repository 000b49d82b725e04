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
        if getattr(lchd, "_deterministic", False):
            self.set_deterministic(True)
        self.use_current_stream()
        self._clouds = []

    def use_current_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        N.check(N.lib().lchd_ctx_set_stream(self._ctx, C.c_void_p(s)))

    def set_deterministic(self, on: bool = True):
        """Pin one sweep kernel family (lchd_ctx_set_deterministic): a pair's score then depends on the pair and the configuration
        only, bit for bit, whatever the batch, the call history or the sharding -- at about half the default throughput."""
        N.check(N.lib().lchd_ctx_set_deterministic(self._ctx, int(bool(on))))

    def enable_timing(self, on: bool = True):
        N.lib().lchd_ctx_enable_timing(self._ctx, int(on))

    def last_ms(self) -> dict:
        return {k: N.lib().lchd_ctx_last_ms(self._ctx, k.encode()) for k in ("cells", "anchors", "env", "sweep")}

    def last_env_points(self) -> int:
        return int(N.lib().lchd_ctx_last_env_points(self._ctx))

    def last_dense_fused(self) -> bool:
        """True if the most recent from_coords call ran the fused sort + sweep kernel (lchd_ctx_last_dense_fused)."""
        return bool(N.lib().lchd_ctx_last_dense_fused(self._ctx))

    def pass_counts(self) -> dict:
        """Passes the context has run: all of them, and the second passes over the pairs of overflowed environments
        (lchd_ctx_pass_count, lchd_ctx_subset_pass_count); store_bytes: environment-store bytes of the last call's passes."""
        lib = N.lib()
        return {"passes": int(lib.lchd_ctx_pass_count(self._ctx)), "subset_passes": int(lib.lchd_ctx_subset_pass_count(self._ctx)),
                "per_pair_passes": int(lib.lchd_ctx_per_pair_pass_count(self._ctx)),
                "store_bytes": int(lib.lchd_ctx_last_store_bytes(self._ctx))}

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

    def from_coords(self, cloud_a, cloud_b, out=None, wf_index=None):
        """LoCoHD.from_coords on two uploaded structures of equal size: pair r = (atom r, atom r), environments = the whole
        structures.  Returns (or fills) a torch float64 CUDA tensor [n]."""
        torch = self.torch
        n = int(N.lib().lchd_cloud_size(cloud_a))
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=torch.device("cuda", self.device))
        assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() >= n
        wf_ptr = None
        if wf_index is not None:
            assert wf_index.is_cuda and wf_index.dtype == torch.int32 and wf_index.is_contiguous() and wf_index.numel() >= n
            wf_ptr = C.c_void_p(wf_index.data_ptr())
        N.check(N.lib().lchd_from_coords_dev(self._ctx, cloud_a, cloud_b, wf_ptr, C.c_void_p(out.data_ptr())))
        return out

    # ---- asynchronous form + trajectory streaming ----------------------------------------------------------------
    def from_primitives_async(self, cloud_a, cloud_b, anchors, threshold_distance: float, out, wf_index=None):
        """Enqueue one pass and return immediately; `finish()` waits for it and raises on errors."""
        torch = self.torch
        assert anchors.is_cuda and anchors.dtype == torch.int64 and anchors.is_contiguous()
        assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() >= anchors.shape[0]
        wf_ptr = None if wf_index is None else C.c_void_p(wf_index.data_ptr())
        N.check(N.lib().lchd_from_primitives_dev_async(self._ctx, cloud_a, cloud_b, C.c_void_p(anchors.data_ptr()), wf_ptr,
                                                       anchors.shape[0], float(threshold_distance), C.c_void_p(out.data_ptr())))

    def finish(self):
        N.check(N.lib().lchd_ctx_finish(self._ctx))

    def frames_buffer(self, template_cloud, capacity_frames: int):
        """Device buffer for `capacity_frames` frames of the template structure (same atoms, new coordinates)."""
        h = C.c_void_p()
        N.check(N.lib().lchd_frames_create(self._ctx, template_cloud, int(capacity_frames), C.byref(h)))
        self._clouds.append(h)
        return h

    def load_frames(self, buf, xyz: np.ndarray, stream=None):
        """Stage xyz [n_frames][n_atoms][3] (host) into a frames buffer; the copy runs on `stream` (a torch.cuda.Stream)
        without waiting, so it overlaps a pass that is running on the session's own stream."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        assert xyz.ndim == 3 and xyz.shape[2] == 3
        sp = None if stream is None else C.c_void_p(stream.cuda_stream)
        N.check(N.lib().lchd_frames_load(self._ctx, buf, N.dp(xyz), xyz.shape[0], sp))

    def set_frame_sources(self, buf, topology):
        """Tell a frames buffer how its primitive atoms are made from SOURCE atoms (a `PrimitiveTopology` from
        PrimitiveAssigner.compile_topology): afterwards `load_atom_frames` takes raw float32 atom coordinates and the
        centroids are evaluated on the device."""
        ss = np.ascontiguousarray(topology.src_start, dtype=np.int32)
        si = np.ascontiguousarray(topology.src_idx, dtype=np.int32)
        N.check(N.lib().lchd_frames_set_sources(self._ctx, buf, N.ip(ss), N.ip(si), int(topology.n_atoms)))

    def load_atom_frames(self, buf, atom_xyz: np.ndarray, stream=None):
        """Stage float32 SOURCE-atom coordinates [n_frames][n_src_atoms][3] (host); H2D copy and the centroid kernel run on
        `stream` without waiting (same overlap rules as load_frames)."""
        atom_xyz = np.ascontiguousarray(atom_xyz, dtype=np.float32)
        assert atom_xyz.ndim == 3 and atom_xyz.shape[2] == 3
        sp = None if stream is None else C.c_void_p(stream.cuda_stream)
        N.check(N.lib().lchd_frames_load_atoms(self._ctx, buf, atom_xyz.ctypes.data_as(C.POINTER(C.c_float)), atom_xyz.shape[0], sp))

    def load_atom_frames_dev(self, buf, atom_xyz, stream=None):
        """As load_atom_frames with the float32 source atoms already on the device: a torch CUDA tensor [n_frames][n_src][3]."""
        assert atom_xyz.is_cuda and atom_xyz.dtype == self.torch.float32 and atom_xyz.is_contiguous() and atom_xyz.dim() == 3
        sp = None if stream is None else C.c_void_p(stream.cuda_stream)
        N.check(N.lib().lchd_frames_load_atoms_dev(self._ctx, buf, C.c_void_p(atom_xyz.data_ptr()), atom_xyz.shape[0], sp))

    def last_convert_ms(self, buf) -> float:
        return float(N.lib().lchd_frames_last_convert_ms(self._ctx, buf))

    def coords_of(self, cloud, n: int) -> np.ndarray:
        """Primitive-atom coordinates currently held by a cloud / frames buffer, as float64 [n][3] on the host."""
        out = np.empty((int(n), 3), dtype=np.float64)
        N.check(N.lib().lchd_cloud_get_coords(self._ctx, cloud, N.dp(out), int(n)))
        return out

    def score_trajectory(self, ref_cloud, frames_xyz: np.ndarray, local_pairs, threshold_distance: float, chunk: int = 1024,
                         topology=None):
        """MD-trajectory mode (python_codes/trajectory_analyzer.py:97-119): score every frame of `frames_xyz`
        [n_frames][n_atoms][3] against the reference structure for the anchor pairs `local_pairs` [(atom in reference,
        atom in frame)].  Frames are streamed in chunks: while chunk k is scored, chunk k+1 is copied on a second
        stream into the other of two buffers.  Returns a float64 array [n_frames][len(local_pairs)].

        With `topology` (PrimitiveAssigner.compile_topology of the trajectory's structure) `frames_xyz` holds the float32
        coordinates of the SOURCE atoms, [n_frames][topology.n_atoms][3], and the per-frame structure -> primitive-atom
        conversion (trajectory_analyzer.py:37-74) runs on the device as well."""
        torch = self.torch
        if topology is not None:
            frames_xyz = np.ascontiguousarray(frames_xyz, dtype=np.float32)
            if frames_xyz.ndim != 3 or frames_xyz.shape[1:] != (topology.n_atoms, 3):
                raise ValueError(f"expected [n_frames][{topology.n_atoms}][3] source-atom coordinates, got {frames_xyz.shape}")
            n_frames, n_atoms = frames_xyz.shape[0], len(topology)
            load = self.load_atom_frames
        else:
            frames_xyz = np.ascontiguousarray(frames_xyz, dtype=np.float64)
            n_frames, n_atoms = frames_xyz.shape[0], frames_xyz.shape[1]
            load = self.load_frames
        lp = np.ascontiguousarray(local_pairs, dtype=np.int64).reshape(-1, 2)
        chunk = max(1, min(int(chunk), n_frames))
        dev = torch.device("cuda", self.device)
        offs = torch.arange(chunk, dtype=torch.int64, device=dev).repeat_interleave(len(lp)) * n_atoms
        anchors = torch.from_numpy(np.tile(lp, (chunk, 1))).to(dev)
        anchors[:, 1] += offs
        anchors = anchors.contiguous()
        out = torch.empty(n_frames * len(lp), dtype=torch.float64, device=dev)
        bufs = [self.frames_buffer(ref_cloud, chunk), self.frames_buffer(ref_cloud, chunk)]
        if topology is not None:
            for b in bufs:
                self.set_frame_sources(b, topology)
        try:
            copy_stream = torch.cuda.Stream(device=dev)
            starts = list(range(0, n_frames, chunk))
            load(bufs[0], frames_xyz[starts[0]:starts[0] + chunk], copy_stream)
            for k, f0 in enumerate(starts):
                nf = min(chunk, n_frames - f0)
                self.from_primitives_async(ref_cloud, bufs[k % 2], anchors[: nf * len(lp)], threshold_distance,
                                           out[f0 * len(lp):(f0 + nf) * len(lp)])
                if k + 1 < len(starts):
                    f1 = starts[k + 1]
                    load(bufs[(k + 1) % 2], frames_xyz[f1:f1 + chunk], copy_stream)
                self.finish()
            return out.cpu().numpy().reshape(n_frames, len(lp))
        finally:  # the two frames buffers go away on every path out (a failed load or finish included)
            try:
                self.finish()
            except Exception:
                pass
            for b in bufs:
                N.lib().lchd_cloud_destroy(self._ctx, b)
                self._clouds.remove(b)

    def close(self):
        if self._ctx:
            from .dist import clear_shard_cache

            clear_shard_cache(self)  # remembered partitions made with this session pin device tensors
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
