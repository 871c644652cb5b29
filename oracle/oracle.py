"""ctypes front-end of oracle/libsaf_oracle.so (TEST INFRASTRUCTURE -- see saf_oracle.c).

Works on CPU torch tensors / numpy arrays.  `OracleVolume` mirrors the buffers of the
reference's ClipFusion / ClipSeemFusion modules so tests can compare buffer by buffer.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

from spatially_aware_ai_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsaf_oracle.so")
_lib = None


def build(force: bool = False):
    src = os.path.join(_HERE, "saf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsaf_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(_SO)
        vp = C.c_void_p
        l.saf_oracle_fuse_frame.restype = C.c_int
        l.saf_oracle_fuse_frame.argtypes = [C.POINTER(_abi.SafVolume), C.POINTER(_abi.SafFrame), vp]
        l.saf_oracle_classify.restype = C.c_int
        l.saf_oracle_classify.argtypes = [C.POINTER(_abi.SafVolume), C.POINTER(_abi.SafFrame), vp, vp]
        l.saf_oracle_backproject_lattice.restype = C.c_int
        l.saf_oracle_backproject_lattice.argtypes = [
            vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_float, vp, vp]
        l.saf_oracle_query_scan.restype = C.c_int
        l.saf_oracle_query_scan.argtypes = [
            vp, C.c_int64, C.c_int64, C.c_int32, vp, C.c_int32, C.c_int64, C.c_int32, C.c_float,
            C.c_int32, vp, vp]
        for name in ("saf_oracle_merge_finalize", "saf_oracle_mean_to_sum"):
            fn = getattr(l, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(_abi.SafVolume), C.c_int64, C.c_int64]
        l.saf_oracle_sample_vertices.restype = C.c_int
        l.saf_oracle_sample_vertices.argtypes = [C.POINTER(_abi.SafVolume), vp, C.c_int64, vp, vp, vp, vp, vp, vp]
        l.saf_oracle_set_threads.argtypes = [C.c_int]
        l.saf_oracle_get_threads.restype = C.c_int
        l.saf_oracle_label_argmax.restype = C.c_int
        l.saf_oracle_label_argmax.argtypes = [vp, C.c_int64, C.c_int32, vp]
        l.saf_oracle_label_components.restype = C.c_int
        l.saf_oracle_label_components.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp,
                                                  C.c_int32, vp, vp, vp]
        _lib = l
    return _lib


def set_threads(n: int):
    """Host threads the fuse restatement spreads x-slabs over (default 1)."""
    lib().saf_oracle_set_threads(int(n))


def _f32(t):
    return torch.as_tensor(t, dtype=torch.float32).contiguous()


class OracleVolume:
    """Volume state on the CPU with the reference's buffer names."""

    def __init__(self, origin, voxel_size, nvox, trunc, feat_dim, n_classes=0, accum_mode=_abi.SAF_RUNNING_MEAN,
                 feat_dtype=torch.float32):
        nvox = torch.as_tensor(nvox)
        self.nvox = nvox
        self.nx, self.ny, self.nz = (int(v) for v in nvox)
        n = self.nx * self.ny * self.nz
        self.n = n
        self.feat_dim = feat_dim
        self.n_classes = n_classes
        self.trunc = float(trunc)
        self.accum_mode = accum_mode
        origin = _f32(origin)
        # identical expression to clipfusion.py:617-622, per axis
        self.axes = [
            (torch.arange(int(nvox[a])) * voxel_size + origin[a]).to(torch.float32).contiguous()
            for a in range(3)
        ]
        self.tsdf = torch.zeros(n)
        self.rgb = torch.zeros(n, 3)
        self.clip_feat = torch.zeros(n, feat_dim, dtype=feat_dtype)
        self.weight = torch.zeros(n, dtype=torch.int32)
        self.tsdf_weight = torch.zeros(n, dtype=torch.int32)
        self.labels_one_hot = torch.zeros(n, n_classes, dtype=torch.int32) if n_classes else None
        self.stats = np.zeros(_abi.SAF_STATS_WORDS, dtype=np.uint64)

    def c_volume(self):
        p = _abi.ptr
        return _abi.SafVolume(
            self.nx, self.ny, self.nz, self.feat_dim, self.n_classes,
            _abi.SAF_BF16 if self.clip_feat.dtype == torch.bfloat16 else _abi.SAF_F32, self.accum_mode,
            self.trunc, p(self.axes[0]), p(self.axes[1]), p(self.axes[2]), p(self.tsdf),
            p(self.tsdf_weight), p(self.weight), p(self.rgb), p(self.clip_feat), p(self.labels_one_hot),
        )

    @staticmethod
    def c_frame(depth, rgb, pose, K, feat, labels=None, rgb_bilinear=False):
        """Single frame (no batch dim).  Returns (struct, keepalive)."""
        depth, rgb, pose, K, feat = (_f32(t) for t in (depth, rgb, pose, K, feat))
        lab = _f32(labels) if labels is not None else None
        h, w = depth.shape
        assert rgb.shape == (h, w, 3) and pose.shape == (4, 4) and K.shape == (3, 3)
        p = _abi.ptr
        fr = _abi.SafFrame(h, w, p(depth), p(rgb), p(pose), p(K), p(feat), feat.shape[1], feat.shape[2],
                           p(lab), int(bool(rgb_bilinear)))
        return fr, (depth, rgb, pose, K, feat, lab)

    def integrate(self, depth_imgs, rgb_imgs, poses, K, feat_maps, label_maps=None, rgb_bilinear=False):
        """Batch semantics of integrate(): frames folded in one after the other."""
        vol = self.c_volume()
        for i in range(len(depth_imgs)):
            fr, keep = self.c_frame(
                depth_imgs[i], rgb_imgs[i], poses[i], K[i], feat_maps[i],
                None if label_maps is None else label_maps[i], rgb_bilinear)
            rc = lib().saf_oracle_fuse_frame(C.byref(vol), C.byref(fr), self.stats.ctypes.data)
            assert rc == 0, rc

    def classify(self, depth, rgb, pose, K, feat):
        vol = self.c_volume()
        fr, keep = self.c_frame(depth, rgb, pose, K, feat)
        mask = np.zeros(self.n, dtype=np.uint8)
        grid = np.zeros((self.n, 2), dtype=np.float32)
        lib().saf_oracle_classify(C.byref(vol), C.byref(fr), mask.ctypes.data, grid.ctypes.data)
        return mask, grid

    def merge_finalize(self):
        vol = self.c_volume()
        assert lib().saf_oracle_merge_finalize(C.byref(vol), 0, self.n) == 0

    def mean_to_sum(self):
        vol = self.c_volume()
        assert lib().saf_oracle_mean_to_sum(C.byref(vol), 0, self.n) == 0


def sample_vertices(vol, verts_index, obj_idx=None, seg_color=None):
    """extract_mesh's vertex sampling on an OracleVolume; returns (feat[V,D], rgb[V,3], obj[V] | None, seg[V,3] | None)."""
    verts = _f32(verts_index)
    nv = verts.shape[0]
    feat = torch.zeros(nv, vol.feat_dim)
    rgb = torch.zeros(nv, 3)
    oi = torch.as_tensor(obj_idx, dtype=torch.int32).contiguous() if obj_idx is not None else None
    sc = _f32(seg_color) if seg_color is not None else None
    oo = torch.zeros(nv) if oi is not None else None
    so = torch.zeros(nv, 3) if sc is not None else None
    c = vol.c_volume()
    p = _abi.ptr
    rc = lib().saf_oracle_sample_vertices(C.byref(c), p(verts), nv, p(feat), p(rgb), p(oi), p(oo), p(sc), p(so))
    assert rc == 0
    return feat, rgb, oo, so


def backproject_lattice(depth, pose, kinv, u_idx, v_idx, max_depth):
    depth, pose, kinv = _f32(depth), _f32(pose), _f32(kinv)
    u = torch.as_tensor(u_idx, dtype=torch.int32).contiguous()
    v = torch.as_tensor(v_idx, dtype=torch.int32).contiguous()
    h, w = depth.shape
    xyz = torch.zeros(len(v) * len(u), 3)
    valid = torch.zeros(len(v) * len(u), dtype=torch.uint8)
    p = _abi.ptr
    rc = lib().saf_oracle_backproject_lattice(p(depth), h, w, p(pose), p(kinv), p(u), len(u), p(v), len(v),
                                              float(max_depth), p(xyz), p(valid))
    assert rc == 0
    return xyz, valid.bool()


def query_scan(feats, text, epilogue, scale=1.0, normalize=False, want_last=False):
    feats, text = _f32(feats), _f32(text)
    n, d = feats.shape
    out = torch.zeros(n, text.shape[0])
    last = torch.zeros(n) if want_last else None
    p = _abi.ptr
    rc = lib().saf_oracle_query_scan(p(feats), n, feats.stride(0), d, p(text), text.shape[0], text.stride(0),
                                     epilogue, float(scale), int(normalize), p(out), p(last))
    assert rc == 0
    return (out, last) if want_last else out


def label_argmax(labels_one_hot):
    lab = torch.as_tensor(labels_one_hot, dtype=torch.int32).contiguous()
    out = torch.zeros(lab.shape[0], dtype=torch.int32)
    assert lib().saf_oracle_label_argmax(_abi.ptr(lab), lab.shape[0], lab.shape[1], _abi.ptr(out)) == 0
    return out


def label_components(labels_grid, null_class=133, min_voxels=3, max_objects=None):
    """flood_fill_3d's object discovery (handy_utils.py:295-480 without a trained in-situ model):
    (voxel_obj_ids int32[nx,ny,nz], first_voxel[k], class[k], count[k])."""
    lab = torch.as_tensor(labels_grid, dtype=torch.int32).contiguous()
    nx, ny, nz = lab.shape
    n = lab.numel()
    mo = n if max_objects is None else max_objects
    ids = torch.empty(n, dtype=torch.int32)
    nobj = torch.zeros(1, dtype=torch.int32)
    first = torch.zeros(mo, dtype=torch.int32)
    cls = torch.zeros(mo, dtype=torch.int32)
    cnt = torch.zeros(mo, dtype=torch.int32)
    rc = lib().saf_oracle_label_components(_abi.ptr(lab), nx, ny, nz, null_class, min_voxels, _abi.ptr(ids), _abi.ptr(nobj),
                                           mo, _abi.ptr(first), _abi.ptr(cls), _abi.ptr(cnt))
    assert rc == 0
    k = min(int(nobj[0]), mo)
    return ids.view(nx, ny, nz), first[:k], cls[:k], cnt[:k]


def wide_scan(feats, text, epilogue="scores", scale=1.0, normalize=True, n_background=0, rescale=False, row_offset=0,
              round_to=None):
    """CPU truth of saf_query_scan_wide_ex's epilogues on top of the oracle's scores (saf_oracle_query_scan with
    SAF_Q_SCORES): the reductions restate, in torch on the CPU, what the reference's callers do with the score matrix --
    vs_background: softmax(scale * [bg..., target])[-1] per target (query_mesh.py:36-39, hypersim_eval.py:76-81);
    row_argmax: first maximum per row (eval_scannet_segmentation.py:553-560); query_max: best row per query.
    ``round_to``: a 16-bit dtype the inputs are rounded to first, as the HIP scan reads them."""
    feats = torch.as_tensor(feats)
    text = torch.as_tensor(text, dtype=torch.float32)[:, : feats.shape[1]]
    if round_to is not None:
        feats, text = feats.to(round_to), text.to(round_to)
    mode = {False: 0, True: 1}.get(normalize, normalize)
    mode = 2 if mode in ("clamp", "clamp_min") else int(mode)
    s = query_scan(feats.float(), text.float(), _abi.SAF_Q_SCORES, scale=scale, normalize=mode)
    if epilogue == "scores":
        return s
    if epilogue == "vs_background":
        bg, tg = s[:, :n_background].double(), s[:, n_background:].double()
        lse = torch.logsumexp(bg, dim=1, keepdim=True)
        p = torch.sigmoid(tg - lse)  # = softmax([bg..., target])[-1]
        if rescale:
            p = ((p - 0.5) * 2).clamp(0, 1)
        return p.float()
    if epilogue == "row_argmax":
        val, idx = s.max(dim=1)
        idx = torch.argmax((s == val[:, None]).int(), dim=1)  # the first maximum
        return idx.int(), val
    if epilogue == "query_max":
        if s.shape[0] == 0:
            return torch.full((s.shape[1],), -float("inf")), torch.full((s.shape[1],), -1, dtype=torch.int64)
        val = s.max(dim=0).values
        row = torch.argmax((s == val[None]).int(), dim=0)
        return val, row.long() + row_offset
    raise ValueError(epilogue)
