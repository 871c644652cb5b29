"""On-disk and wire formats of the fused results, behind include/saf.h's saf_save_npy /
saf_mesh_json / saf_save_ply.

The reference writes its artefacts with np.save and trimesh and answers queries with json.dumps of Python lists
(clip_seem_fusion.py:553-607, handy_utils.py:214-241); the formats stay exactly those -- `np.load`, `trimesh.load_mesh`
/ open3d and any JSON parser read them (query_mesh.py:21-25, :42; handy_utils.py:219) -- only the writers change: device
arrays stream to disk through pinned buffers without a host copy of the whole array, and meshes are serialised natively.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import abc as _abc

import numpy as np
import torch

from ._lib import SafError, check, lib

_NPY_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.int32: 3, torch.int64: 4, torch.uint8: 5}


def save_npy(path, tensor):
    """``np.save(path, tensor)`` for a torch tensor on the HIP device (or the CPU): C-order .npy v1.0.  bfloat16 has no
    NumPy dtype and is written as 2-byte records.  Like np.save, ``.npy`` is appended when the path lacks it."""
    path = os.fspath(path)
    if not path.endswith(".npy"):
        path += ".npy"
    t = tensor.detach()
    if t.dtype not in _NPY_CODE:
        raise SafError(f"save_npy: unsupported dtype {t.dtype}")
    if not t.is_contiguous():
        t = t.contiguous()
    shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
    stream = torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None
    if t.is_cuda:
        with torch.cuda.device(t.device):
            rc = lib().saf_save_npy(t.data_ptr(), 1, _NPY_CODE[t.dtype], shape, t.dim(), path.encode(), stream)
    else:
        rc = lib().saf_save_npy(t.data_ptr(), 0, _NPY_CODE[t.dtype], shape, t.dim(), path.encode(), None)
    check(rc, "saf_save_npy")
    return path


def _host(a, dtype):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


def mesh_to_json(verts, faces, colors=None) -> bytes:
    """The JSON the reference sends to its clients -- ``{"vertices": ..., "faces": ..., "colors": ...}`` with nested lists
    (clip_seem_fusion.py:553-559, handy_utils.py:233-239) -- as UTF-8 bytes, serialised natively.  ``json.loads`` of the
    result equals ``{"vertices": verts.tolist(), "faces": faces.tolist(), "colors": colors.tolist()}`` for f32 inputs."""
    v = _host(verts, np.float32).reshape(-1, 3)
    f = _host(faces, np.int32).reshape(-1, 3) if faces is not None else np.zeros((0, 3), np.int32)
    c = None
    if colors is not None:
        c = _host(colors, np.float32)
        c = c.reshape(len(v), c.shape[-1] if c.ndim > 1 else 1)
    out, n = C.c_void_p(), C.c_int64()
    rc = lib().saf_mesh_json(v.ctypes.data if len(v) else None, len(v), f.ctypes.data if len(f) else None, len(f),
                             None if c is None else c.ctypes.data, 0 if c is None else max(1, c.shape[1]), C.byref(out),
                             C.byref(n))
    check(rc, "saf_mesh_json")
    try:
        return C.string_at(out, n.value)
    finally:
        lib().saf_free(out)


class ArrayList(_abc.Sequence):
    """A NumPy array standing in for the nested Python list the reference's data model holds in its place -- the voxel
    coordinates of an object (``flood_fill_3d`` appends tuples, handy_utils.py:430-452) and the per-object meshes
    (``.tolist()`` of every vertex, face and colour array, clip_seem_fusion.py:393-417): hundreds of thousands of small
    Python objects per scan, seconds of interpreter time, only to be turned into JSON text or back into an array by whoever
    reads them.  Reads like the list (``len``, indexing, iteration, ``==`` against lists), converts with ``tolist()`` or
    ``np.asarray``, and is written to JSON natively (``dumps_scene_knowledge``)."""

    __slots__ = ("array", "tuples")

    def __init__(self, array, tuples=False):
        self.array = np.ascontiguousarray(array)
        self.tuples = bool(tuples)  # rows as tuples (the voxel lists) or as lists (tolist())

    def __len__(self):
        return len(self.array)

    def __getitem__(self, i):
        v = self.array[i]
        if isinstance(i, slice):
            return ArrayList(v, self.tuples)
        v = v.tolist()
        return tuple(v) if self.tuples and isinstance(v, list) else v

    def __iter__(self):
        rows = self.array.tolist()
        return iter(map(tuple, rows)) if self.tuples and self.array.ndim > 1 else iter(rows)

    def __array__(self, dtype=None, copy=None):
        return self.array if dtype is None else self.array.astype(dtype)

    def tolist(self):
        rows = self.array.tolist()
        return list(map(tuple, rows)) if self.tuples and self.array.ndim > 1 else rows

    def __eq__(self, other):
        if isinstance(other, ArrayList):
            return self.array.shape == other.array.shape and bool((self.array == other.array).all())
        if isinstance(other, (list, tuple)):
            return len(other) == len(self) and self.tolist() == [tuple(r) if self.tuples and isinstance(r, (list, tuple)) else r for r in other]
        return NotImplemented

    def __repr__(self):
        return f"ArrayList(shape={tuple(self.array.shape)}, dtype={self.array.dtype})"


_JSON_CODE = {np.dtype(np.float32): 0, np.dtype(np.int32): 3, np.dtype(np.int64): 4, np.dtype(np.float64): 6}


def array_to_json(a) -> str:
    """The JSON text of ``a.tolist()`` for a 1-D or 2-D host array (f32 / f64 / i32 / i64), rendered natively: parses to
    exactly what ``json.loads(json.dumps(a.tolist()))`` gives."""
    a = np.ascontiguousarray(a.array if isinstance(a, ArrayList) else a)
    if a.dtype not in _JSON_CODE:
        a = a.astype(np.float64 if a.dtype.kind == "f" else np.int64)
    if a.ndim > 2:
        a = a.reshape(a.shape[0], -1)
    rows, cols = (a.shape[0], 0) if a.ndim == 1 else a.shape
    if a.ndim == 2 and cols == 0:
        return "[" + ", ".join("[]" for _ in range(rows)) + "]"
    out, n = C.c_void_p(), C.c_int64()
    check(lib().saf_array_json(a.ctypes.data if a.size else None, _JSON_CODE[a.dtype], rows, cols, C.byref(out), C.byref(n)), "saf_array_json")
    try:
        return C.string_at(out, n.value).decode()
    finally:
        lib().saf_free(out)


def dumps_scene_knowledge(obj) -> str:
    """``json.dumps(obj, default=str)`` (what the reference writes to scene_knowledge.json, clip_seem_fusion.py:603-604) for an
    object whose bulky lists are ``ArrayList``s: the small skeleton goes through the standard encoder, every array is rendered
    natively and spliced in.  The text parses to the same object as the reference's."""
    import json

    arrays = []

    def strip(o):
        if isinstance(o, ArrayList):
            arrays.append(o)
            return f"\u0000saf-array-{len(arrays) - 1}\u0000"
        if isinstance(o, dict):
            return {k: strip(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [strip(v) for v in o]
        return o

    text = json.dumps(strip(obj), default=str)
    if not arrays:
        return text
    parts = text.split('"\\u0000saf-array-')
    out = [parts[0]]
    for p in parts[1:]:
        k, rest = p.split('\\u0000"', 1)
        out.append(array_to_json(arrays[int(k)]))
        out.append(rest)
    return "".join(out)


def save_ply(path, verts, faces, vertex_colors=None):
    """Binary PLY with per-vertex colours, the layout of the reference's mesh_rgb.ply / mesh_segmentation.ply
    (trimesh export, clip_seem_fusion.py:584-600).  ``vertex_colors`` [V,3|4] floats in 0..1."""
    v = _host(verts, np.float32).reshape(-1, 3)
    f = _host(faces, np.int32).reshape(-1, 3)
    c = None if vertex_colors is None else _host(vertex_colors, np.float32).reshape(len(v), -1)
    rc = lib().saf_save_ply(os.fspath(path).encode(), v.ctypes.data, len(v), f.ctypes.data if len(f) else None, len(f),
                            None if c is None else c.ctypes.data, 0 if c is None else c.shape[1])
    check(rc, "saf_save_ply")
    return path


def save_scene_arrays(out_dir, fusion, vert_clip_feat=None, vertex_obj_idx=None):
    """The numpy artefacts of save_files_and_broadcast (clip_seem_fusion.py:566-581) straight from the device:
    voxel_rgb.npy [nx,ny,nz,3], voxel_clip_feats.npy [nx,ny,nz,D] (clip_seem_fusion.py:335-338 reshapes the flat buffers
    this way), and the vertex arrays when given."""
    os.makedirs(out_dir, exist_ok=True)
    nx, ny, nz = (int(v) for v in fusion.nvox)
    paths = {"voxel_rgb": save_npy(os.path.join(out_dir, "voxel_rgb.npy"), fusion.rgb.view(nx, ny, nz, 3)),
             "voxel_clip_feats": save_npy(os.path.join(out_dir, "voxel_clip_feats.npy"),
                                          fusion.clip_feat.view(nx, ny, nz, -1))}
    if vert_clip_feat is not None:
        paths["vertex_clip_feats"] = save_npy(os.path.join(out_dir, "vertex_clip_feats.npy"), torch.as_tensor(vert_clip_feat))
    if vertex_obj_idx is not None:
        paths["vertex_obj_idx"] = save_npy(os.path.join(out_dir, "vertex_obj_idx.npy"), torch.as_tensor(vertex_obj_idx))
    return paths
