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
