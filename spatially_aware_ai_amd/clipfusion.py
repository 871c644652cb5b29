"""Host-side mirror of the reference's ``clipfusion.py`` hot-path API on MI355X.

Same names, arguments and error behaviour as the reference classes so that its callers
(`run_clipfusion`, `InSituManager`, `query_mesh.py`, the eval scripts) can switch imports:

  * ``ClipFusion``      -- reference clipfusion.py:575-763 (volume + ``integrate``)
  * ``Clip``            -- reference clipfusion.py:766-1039 (tiled image features, text query head)
  * ``backproject_pcd`` -- reference clipfusion.py:510-572
  * ``get_pix_vecs``    -- reference clipfusion.py:497-507
  * ``scene_bounds``    -- the bounds arithmetic of clipfusion.py:1098-1106

PyTorch is plumbing here (device memory, streams, the backbone GEMMs); the fusion loop and the
query scan are the hand-written HIP kernels behind ``include/saf.h``.  There is no CPU
fallback: tensors that are not on the HIP device raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _abi
from ._lib import SafError, check, current_stream_ptr, lib, require_cuda

_FRAME_WORDS = C.sizeof(_abi.SafFrame) // 8
assert C.sizeof(_abi.SafFrame) == 72 and _abi.SafFrame.depth.offset == 8 and _abi.SafFrame.npy.offset == 48 and \
    _abi.SafFrame.label_map.offset == 56 and _abi.SafFrame.rgb_bilinear.offset == 64, "struct saf_frame layout (include/saf.h)"

# --------------------------------------------------------------------------------------------
# volume plumbing shared by ClipFusion and ClipSeemFusion
# --------------------------------------------------------------------------------------------


def _axis_tables(origin, voxel_size, nvox, index_offset=(0, 0, 0), x_planes=None):
    """Per-axis voxel-centre coordinates, element for element what the reference computes as
    ``xyz_idx * voxel_size + origin`` (clipfusion.py:617-622): int64 index * python float -> f32
    product, + f32 origin.  ``index_offset``: the module holds the sub-grid that starts at this voxel index of the grid
    anchored at ``origin`` (a slab of a voxel-sharded volume): the same expression on the same indices, bit for bit.
    ``x_planes``: the x indices the module holds, in its own order (a slab made of several blocks of x-planes)."""
    origin = torch.as_tensor(origin).detach().cpu()
    idx = [torch.arange(int(nvox[a])) + int(index_offset[a]) for a in range(3)]
    if x_planes is not None:
        idx[0] = torch.as_tensor(x_planes, dtype=torch.int64).detach().cpu()
        if idx[0].dim() != 1 or idx[0].numel() != int(nvox[0]):
            raise ValueError("x_planes must list nvox[0] x indices")
        # The windowed path classifies 4 x 4 x 16 bricks against a bounding sphere built from a brick's first and last
        # x-plane: every group of 4 consecutive entries must be 4 consecutive planes (a rank's slab is made of whole blocks).
        if int(nvox[0]) % 4 == 0:
            q = idx[0].view(-1, 4)
            if not bool((q[:, 1:] - q[:, :-1] == 1).all()):
                raise ValueError("x_planes must consist of runs of 4 consecutive x indices (blocks of slab_planes_of_rank)")
    return [(idx[a] * voxel_size + origin[a]).to(torch.float32).contiguous() for a in range(3)]


# `fusion.borrow_inputs = True` (default: SAF_BORROW_INPUTS=1 in the environment, else False): the queue behind integrate() does
# not copy a call's depth / rgb / label images into its staging ring but reads them where they are when their window is fused --
# up to 512 frames later.  The caller promises not to WRITE those tensors until flush() (dropping them is fine: the queue holds
# them); the reference's loop (a fresh batch per DataLoader step, clip_seem_fusion.py:303-313) keeps that promise by construction.
_BORROW_DEFAULT = os.environ.get("SAF_BORROW_INPUTS", "0") == "1"
_VOLUME_BUFFERS = frozenset(("tsdf", "rgb", "clip_feat", "weight", "tsdf_weight", "labels_one_hot", "fuse_stats"))


class _FusionVolumeMixin:
    """Buffers, workspace and the C-ABI call shared by both fusion modules."""

    def _init_volume(self, origin, voxel_size, nvox, trunc, feat_dim, n_classes=0, keep_xyz_world=True,
                     feat_dtype=torch.float32, index_offset=(0, 0, 0), x_planes=None, device=None):
        """``device`` (not in the reference): create the buffers there instead of on the host.  The reference builds its module on
        the host and moves it (clipfusion.py:1119, clip_seem_fusion.py:291-302) -- gigabytes of zeros through pageable memory
        (0.7 s at 127 x 104 x 116 x 512); the axis tables are computed on the host either way."""
        zeros = lambda *a, **k: torch.zeros(*a, device=device, **k)
        nvox = torch.as_tensor(nvox)
        n = int(torch.prod(nvox.long()))
        self.origin = origin
        self.voxel_size = voxel_size
        self.nvox = nvox
        self.trunc = trunc
        self.n_clip_feats = feat_dim
        self.accum_mode = _abi.SAF_RUNNING_MEAN
        self.register_buffer("tsdf", zeros(n, dtype=torch.float32))
        self.register_buffer("rgb", zeros((n, 3), dtype=torch.float32))
        if feat_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("feat_dtype must be torch.float32 (the reference layout) or torch.bfloat16")
        self.register_buffer("clip_feat", zeros((n, feat_dim), dtype=feat_dtype))
        self.register_buffer("weight", zeros(n, dtype=torch.int32))
        self.register_buffer("tsdf_weight", zeros(n, dtype=torch.int32))
        if n_classes:
            self.register_buffer("labels_one_hot", zeros((n, n_classes), dtype=torch.int32))
        self.index_offset = tuple(int(v) for v in index_offset)
        self.x_planes = None if x_planes is None else torch.as_tensor(x_planes, dtype=torch.int64).detach().cpu().clone()
        ax = _axis_tables(origin, voxel_size, nvox, self.index_offset, self.x_planes)
        if device is not None:
            ax = [t.to(device) for t in ax]
        # not in the reference's state_dict: derived tables the sweep kernel reads instead of xyz_world
        self.register_buffer("axis_x", ax[0], persistent=False)
        self.register_buffer("axis_y", ax[1], persistent=False)
        self.register_buffer("axis_z", ax[2], persistent=False)
        self.register_buffer("fuse_stats", zeros(_abi.SAF_STATS_WORDS, dtype=torch.int64), persistent=False)
        if keep_xyz_world:
            nx, ny, nz = (int(v) for v in nvox)
            xyz_world = torch.stack(
                (
                    ax[0][:, None, None].expand(nx, ny, nz),
                    ax[1][None, :, None].expand(nx, ny, nz),
                    ax[2][None, None, :].expand(nx, ny, nz),
                ),
                dim=-1,
            ).reshape(-1, 3)
            self.register_buffer("xyz_world", xyz_world)
        self._workspace = None
        self._shard_stripes = None  # [(first, count)] while the volume holds only its reduce-scattered stripes of a merged job
        self.__dict__.setdefault("defer_frames", True)  # queue small integrate() calls into 128-frame windows
        self.__dict__["_pending_n"] = 0
        self.__dict__["_stage"] = None
        self.__dict__["_feat_stale"] = False
        self.__dict__["_session"] = None       # saf_fuse_session handle of the queue's flushes (created at the first one)
        self.__dict__["_session_open"] = False  # a pushed window's row kernel is still owed (saf_fuse_session_finish)
        self.__dict__["_fs"] = None             # the queue's own stream: a flush runs beside the staging of later frames
        self.__dict__["_fs_event"] = None       # recorded behind the last finish: whoever reads the volume waits for it

    def __del__(self):
        h = self.__dict__.get("_session")
        if h:
            try:
                lib().saf_fuse_session_destroy(h)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass

    # -- C structs -------------------------------------------------------------------------
    def _c_volume(self, for_fuse=False):
        """The saf_volume descriptor of the buffers.  ``for_fuse``: for the fuse kernels themselves -- neither the
        queued frames nor the deferred clear (reset) are resolved first."""
        if not for_fuse:
            self._sync_volume()
        b = self._buffers
        nx, ny, nz = (int(v) for v in self.nvox)
        labels = b.get("labels_one_hot")
        require_cuda(b["clip_feat"], "the fusion volume")
        for name in ("tsdf", "rgb", "clip_feat", "weight", "tsdf_weight"):
            if not b[name].is_contiguous():
                raise SafError(f"buffer {name} must be contiguous")
        p = _abi.ptr
        return _abi.SafVolume(
            nx, ny, nz, int(self.n_clip_feats), 0 if labels is None else int(labels.shape[1]),
            _abi.SAF_BF16 if b["clip_feat"].dtype == torch.bfloat16 else _abi.SAF_F32, int(self.accum_mode),
            float(self.trunc),
            p(b["axis_x"]), p(b["axis_y"]), p(b["axis_z"]),
            p(b["tsdf"]), p(b["tsdf_weight"]), p(b["weight"]), p(b["rgb"]), p(b["clip_feat"]), p(labels),
        )

    def _get_workspace(self, npy, npx, hw=None):
        """``hw`` = (height, width) of the frames: a volume of a million voxels or more also gets room for the windowed path's
        tiled depth copies (0.63 GB at 640 x 480; half the classification's cache lines); the library falls back to
        the frames' own images when the room is not there -- results are identical either way."""
        tsdf = self._buffers["tsdf"]
        n = tsdf.numel()
        # for THIS volume (width, dtype, SAF_WIN_FORM): the brick form's 6.5 GB of segment pools only where it would run
        # (SAF_TILED_MIN_VOXELS, read per call: the tests ask for the tiled copies on small volumes)
        if hw is not None and n >= int(os.environ.get("SAF_TILED_MIN_VOXELS", 1 << 20)):
            need = lib().saf_fuse_workspace_bytes_for_frames(C.byref(self._c_volume(for_fuse=True)), int(npy), int(npx),
                                                             int(hw[0]), int(hw[1]))
        else:
            need = lib().saf_fuse_workspace_bytes_for(C.byref(self._c_volume(for_fuse=True)), int(npy), int(npx))
        ws = self._workspace
        if ws is None or ws.numel() < need or ws.device != tsdf.device:
            ws = torch.empty(need, dtype=torch.uint8, device=tsdf.device)
            self._workspace = ws
        return ws

    @staticmethod
    def _f32c(t, name):
        require_cuda(t, name)
        if t.dtype != torch.float32:
            t = t.float()
        return t.contiguous()

    def _make_frames(self, depth_imgs, rgb_imgs, poses, K, clip_feat_img, label_maps, rgb_bilinear, borrowed=None):
        """C descriptors for a batch; returns (ctypes array, keepalive list, npy, npx).  ``borrowed`` ([B, 3] int64, or None):
        per frame the addresses of a caller's depth / rgb / label images that stand in for the batch's (0: none)."""
        depth_imgs = self._f32c(depth_imgs, "depth_imgs")
        rgb_imgs = self._f32c(rgb_imgs, "rgb_imgs")
        poses = self._f32c(poses, "poses")
        K = self._f32c(K, "K")
        feat = self._f32c(clip_feat_img, "clip feature map")
        bsz, h, w = depth_imgs.shape
        if rgb_imgs.shape != (bsz, h, w, 3):
            raise ValueError(f"rgb_imgs must be [B,H,W,3], got {tuple(rgb_imgs.shape)}")
        if poses.shape != (bsz, 4, 4) or K.shape != (bsz, 3, 3):
            raise ValueError("poses must be [B,4,4] and K [B,3,3]")
        if feat.dim() != 4 or feat.shape[0] != bsz or feat.shape[1] < self.n_clip_feats:
            raise ValueError(f"feature map must be [B,D>={self.n_clip_feats},npy,npx], got {tuple(feat.shape)}")
        labs = lab_ptrs = None
        if torch.is_tensor(label_maps):  # one [B,H,W] tensor (the staging ring)
            labs = self._f32c(label_maps, "label maps")
            if labs.shape != (bsz, h, w):
                raise ValueError("label maps must be [B,H,W]")
            lab_ptrs = labs.data_ptr() + np.arange(bsz, dtype=np.int64) * (4 * h * w)
            labs = [labs]
        elif label_maps is not None:
            labs = [self._f32c(m, "label map") for m in label_maps]
            for m in labs:
                if m.shape != (h, w):
                    raise ValueError("label map must be [H,W]")
            lab_ptrs = np.fromiter((m.data_ptr() for m in labs), dtype=np.int64, count=bsz)
        npy, npx = int(feat.shape[2]), int(feat.shape[3])
        # the descriptors as int64 words (struct saf_frame, include/saf.h: 72 bytes), filled column by column: a Python
        # object per frame and field would be thousands of short-lived objects per flush -- garbage-collector pauses
        # in the middle of the one-frame-per-call loop
        arr = np.zeros((bsz, _FRAME_WORDS), dtype=np.int64)
        idx = np.arange(bsz, dtype=np.int64)
        arr[:, 0] = h | (w << 32)
        arr[:, 1] = depth_imgs.data_ptr() + idx * (4 * h * w)
        arr[:, 2] = rgb_imgs.data_ptr() + idx * (12 * h * w)
        arr[:, 3] = poses.data_ptr() + idx * 64
        arr[:, 4] = K.data_ptr() + idx * 36
        arr[:, 5] = feat.data_ptr() + idx * (4 * int(feat.shape[1]) * npy * npx)
        arr[:, 6] = npy | (npx << 32)
        if lab_ptrs is not None:
            arr[:, 7] = lab_ptrs
        arr[:, 8] = int(bool(rgb_bilinear))
        if borrowed is not None:
            for col, k in ((1, 0), (2, 1), (7, 2)):
                arr[:, col] = np.where(borrowed[:, k] != 0, borrowed[:, k], arr[:, col])
        return (_abi.SafFrame * bsz).from_buffer(arr), (depth_imgs, rgb_imgs, poses, K, feat, labs), npy, npx

    # -- the deferred window queue ---------------------------------------------------------------
    # The reference calls integrate() with ONE frame per DataLoader batch (clip_seem_fusion.py:303-313,
    # clipfusion.py:1120-1133).  One frame per C call would run the per-frame pipeline; the windowed path
    # (saf_fuse_frames with 16+ frames: every touched feature row travels to HBM once per 128-frame window)
    # needs many frames in one call.  Small calls are therefore queued -- their inputs copied into a staging ring
    # of four windows (512 slots) -- and handed to the library 32 at a time (round 6): every 32 staged frames are pushed into a
    # streaming session (saf_fuse_session_push) on a stream of the queue's own -- one classification launch, which runs beside
    # the row kernel of the window before; the 128th frame of a window brings its row kernel -- and the caller's stream goes on
    # staging the next frames into the ring's other quarters meanwhile (rounds 2-5: one saf_fuse_frames call per flush on the
    # caller's stream -- every flush exposed its first window's classification, 3.7 ms, could not start before the window's
    # last frame was staged, 3 ms, and later frames' staging kernels queued behind the flush, 3 ms: 0.86-0.90 of the bulk rate).  Whatever reads or replaces the volume -- the registered buffers (attribute access, state_dict, .to()), stats(),
    # extract_mesh, the merge -- pushes what is staged, finishes the session and lets its stream wait for it.
    # The device paths are bit-identical, so a caller cannot tell -- except by speed.
    _DEFER_MAX_BATCH = 15  # calls of 16+ frames take the windowed path by themselves
    # up to four windows per flush (from the second one on, a window is classified beside its predecessor's rows); a full
    # window is flushed earlier when the device has finished the previous flush and would otherwise idle
    _QUEUE_FRAMES = 4 * _abi.SAF_WINDOW_FRAMES
    _PUSH_FRAMES = 32  # frames of one classification launch (one mask plane of a window): the unit the session takes

    def _defer_ok(self, bsz, npy, npx):
        if not self.__dict__.get("defer_frames", True) or bsz > self._DEFER_MAX_BATCH:
            return False
        d = int(self.n_clip_feats)
        bf16 = self._buffers["clip_feat"].dtype == torch.bfloat16
        # the shapes saf_fuse_frames takes on the windowed path (include/saf.h); others gain nothing from a queue
        return d <= 1024 and d % (512 if bf16 else 256) == 0 and npy + 3 <= 255 and npx + 3 <= 255

    def _fuse(self, depth_imgs, rgb_imgs, poses, K, clip_feat_img, label_maps=None, rgb_bilinear=False, lazy_feat=None):
        """``lazy_feat`` = (fn, (C, npy, npx)) instead of ``clip_feat_img``: the feature maps of these frames are
        ``fn(rgb[n,H,W,3]) -> [n,C,npy,npx]`` and may be computed later -- when the queue is flushed, for all queued frames in
        one backbone batch (the reference feeds its ViT 35 tiles per call; a flush feeds it 128 x 35)."""
        bsz = int(depth_imgs.shape[0])
        f32 = torch.float32
        if clip_feat_img is None:
            fn, fshape = lazy_feat
            lazy_ok = self._defer_ok(bsz, fshape[1], fshape[2]) and all(
                t.is_cuda and t.dtype == f32 and t.is_contiguous() for t in (depth_imgs, rgb_imgs, poses, K)) and (
                label_maps is None or all(m.is_cuda and m.dtype == f32 and m.is_contiguous() for m in label_maps))
            if not lazy_ok:
                return self._fuse(depth_imgs, rgb_imgs, poses, K, fn(rgb_imgs), label_maps, rgb_bilinear)
            # the backbone runs at flush time: under the autocast state of THIS call
            lazy = (fn, bool(torch.is_autocast_enabled("cuda")), torch.get_autocast_dtype("cuda"))
        else:
            lazy = None
            if clip_feat_img.dim() != 4 or not self._defer_ok(bsz, int(clip_feat_img.shape[2]), int(clip_feat_img.shape[3])):
                self._flush_pending(final=True)  # (frames are fused in call order)
                return self._fuse_now(depth_imgs, rgb_imgs, poses, K, clip_feat_img, label_maps, rgb_bilinear)
            fshape = tuple(int(v) for v in clip_feat_img.shape[1:])
        h, w = int(depth_imgs.shape[1]), int(depth_imgs.shape[2])
        if tuple(rgb_imgs.shape) != (bsz, h, w, 3):
            raise ValueError(f"rgb_imgs must be [B,H,W,3], got {tuple(rgb_imgs.shape)}")
        if tuple(poses.shape) != (bsz, 4, 4) or tuple(K.shape) != (bsz, 3, 3):
            raise ValueError("poses must be [B,4,4] and K [B,3,3]")
        if lazy is None and (clip_feat_img.shape[0] != bsz or clip_feat_img.shape[1] < self.n_clip_feats):
            raise ValueError(f"feature map must be [B,D>={self.n_clip_feats},npy,npx], got {tuple(clip_feat_img.shape)}")
        if fshape[0] < self.n_clip_feats:
            raise ValueError(f"the backbone yields {fshape[0]} channels, the volume holds {self.n_clip_feats}")
        for t, name in ((depth_imgs, "depth_imgs"), (rgb_imgs, "rgb_imgs"), (poses, "poses"), (K, "K")) + (
                () if lazy is not None else ((clip_feat_img, "clip feature map"),)):
            require_cuda(t, name)
        if label_maps is not None:
            for m in label_maps:
                require_cuda(m, "label map")
                if tuple(m.shape) != (h, w):
                    raise ValueError("label map must be [H,W]")
        # frames queued together share the shapes and, for deferred features, the backbone call and its autocast state
        key = (h, w, tuple(fshape), label_maps is not None, bool(rgb_bilinear), None if lazy is None else (id(lazy[0]),) + lazy[1:])
        st = self.__dict__.get("_stage")
        if st is None or st["key"] != key:
            self._flush_pending(final=True)  # (the old ring's frames; the session of the old shape)
            dev = self._buffers["tsdf"].device
            n = self._QUEUE_FRAMES
            mk = lambda *shape: torch.empty((n,) + shape, dtype=torch.float32, device=dev)
            st = {"key": key, "depth": mk(h, w), "rgb": mk(h, w, 3), "pose": mk(4, 4), "K": mk(3, 3),
                  "feat": mk(*key[2]), "labels": mk(h, w) if label_maps is not None else None, "lazy": lazy,
                  # the frames staged and not yet handed over lie in slots [ring0, ring0 + _pending_n); free_ev[q]: the event
                  # behind the row kernel that reads quarter q's frames (None: free); held: quarters whose row kernel is owed
                  "ring0": 0, "free_ev": [None] * (self._QUEUE_FRAMES // _abi.SAF_WINDOW_FRAMES), "sess_frames": 0,
                  # borrow_inputs: per slot the addresses of the caller's depth / rgb / label images that were NOT copied (0: the
                  # ring's) and the tensors themselves, held until the slot is staged again (behind its row kernel's event)
                  "bptr": np.zeros((n, 3), dtype=np.int64), "brefs": [None] * n,
                  "src": _abi.SafFrame(h, w, None, None, None, None, None, key[2][1], key[2][2], None, 0),
                  "dst": _abi.SafFrame(h, w, None, None, None, None, None, key[2][1], key[2][2], None, 0)}
            # base addresses and byte strides of the ring's slots (no tensor views per call)
            st["base"] = tuple(st[k].data_ptr() if st[k] is not None else 0 for k in ("depth", "rgb", "pose", "K", "feat", "labels"))
            st["step"] = (4 * h * w, 12 * h * w, 64, 36, 4 * key[2][0] * key[2][1] * key[2][2], 4 * h * w)
            self.__dict__["_stage"] = st
        dev = self._buffers["tsdf"].device
        with torch.cuda.device(dev):
            # current_stream(dev), not current_stream(): without a device torch asks is_available() first, which looks up
            # an environment variable by raising and catching a KeyError -- up to 100 us per call in a long-lived process
            stream = torch.cuda.current_stream(dev)
            fast = lazy is not None or (
                all(t.dtype == f32 for t in (depth_imgs, rgb_imgs, poses, K, clip_feat_img)) and depth_imgs.is_contiguous()
                and rgb_imgs.is_contiguous() and poses.is_contiguous() and K.is_contiguous() and
                (label_maps is None or all(m.dtype == f32 and m.is_contiguous() for m in label_maps)))
            src, dst, base, step = st["src"], st["dst"], st["base"], st["step"]
            # (a deferred backbone reads the queued frames' rgb from the ring: nothing to borrow there)
            borrow = fast and lazy is None and bool(self.__dict__.get("borrow_inputs", _BORROW_DEFAULT))
            raw_stream = stream.cuda_stream
            win = _abi.SAF_WINDOW_FRAMES
            if fast:
                fs = (0, 0, 0, 0) if lazy is not None else clip_feat_img.stride()
                sp = (depth_imgs.data_ptr(), rgb_imgs.data_ptr(), poses.data_ptr(), K.data_ptr(),
                      0 if lazy is not None else clip_feat_img.data_ptr())
            for i in range(bsz):
                k = self.__dict__["_pending_n"]
                if st["ring0"] + k >= self._QUEUE_FRAMES:
                    # a full ring means an earlier flush failed and kept its frames: try again (it re-raises) -- never
                    # stage into a slot beyond the ring
                    self._flush_pending()
                    k = self.__dict__["_pending_n"]
                    if st["ring0"] + k >= self._QUEUE_FRAMES:
                        raise SafError("the staging ring is full and could not be flushed")
                slot = st["ring0"] + k
                if slot % win == 0:  # a quarter of the ring is reused: the row kernel that read its frames must be done
                    ev = st["free_ev"][slot // win]
                    if ev is not None:
                        stream.wait_event(ev)
                        st["free_ev"][slot // win] = None
                if fast:  # one launch per frame (saf_stage_frame); addresses by arithmetic: no tensor views, no new descriptors
                    src.depth, src.rgb, src.pose = sp[0] + i * step[0], sp[1] + i * step[1], sp[2] + i * 64
                    src.K, src.feat_map = sp[3] + i * 36, (None if lazy is not None else sp[4] + i * fs[0] * 4)
                    dst.depth, dst.rgb, dst.pose = base[0] + slot * step[0], base[1] + slot * step[1], base[2] + slot * step[2]
                    dst.K, dst.feat_map = base[3] + slot * step[3], base[4] + slot * step[4]
                    if label_maps is not None:
                        src.label_map, dst.label_map = label_maps[i].data_ptr(), base[5] + slot * step[5]
                    if borrow:  # the images stay where the caller has them: 5 MB per frame that are not copied
                        st["bptr"][slot] = (src.depth, src.rgb, src.label_map or 0)
                        st["brefs"][slot] = (depth_imgs, rgb_imgs, None if label_maps is None else label_maps[i])
                        src.depth = src.rgb = src.label_map = dst.depth = dst.rgb = dst.label_map = None
                    elif st["brefs"][slot] is not None:
                        st["bptr"][slot] = 0
                        st["brefs"][slot] = None
                    check(lib().saf_stage_frame(C.byref(src), key[2][0], fs[1], fs[2], fs[3], C.byref(dst), raw_stream),
                          "saf_stage_frame")
                else:  # other dtypes / layouts: PyTorch copies convert
                    if st["brefs"][slot] is not None:
                        st["bptr"][slot] = 0
                        st["brefs"][slot] = None
                    st["depth"][slot].copy_(depth_imgs[i], non_blocking=True)
                    st["rgb"][slot].copy_(rgb_imgs[i], non_blocking=True)
                    st["pose"][slot].copy_(poses[i], non_blocking=True)
                    st["K"][slot].copy_(K[i], non_blocking=True)
                    st["feat"][slot].copy_(clip_feat_img[i], non_blocking=True)
                    if label_maps is not None:
                        st["labels"][slot].copy_(label_maps[i], non_blocking=True)
                self.__dict__["_pending_n"] = k + 1
                if (slot + 1) % self._PUSH_FRAMES == 0:
                    # one classification launch's worth of frames is staged: their depth tiles are computed now, behind the staging
                    # on this stream, and the chunk BEFORE them is handed over -- its staging and tiles have completed meanwhile
                    # (the host needs 0.7 ms to stage 32 frames), so its launch is queued with no cross-stream wait in front
                    # (a session's FIRST window is pushed chunk by chunk at once, each launch waiting for its frames: nothing runs
                    #  beside it yet, a barrier in front of a launch costs nothing there, and the scan starts 0.7 ms earlier)
                    self._prepare_chunk(st, slot + 1 - self._PUSH_FRAMES, self._PUSH_FRAMES, stream)
                    if st["sess_frames"] < win:
                        self._flush_pending()
                    elif self.__dict__["_pending_n"] > self._PUSH_FRAMES:
                        self._flush_pending(keep=self._PUSH_FRAMES)
            if fast:  # the sources are read asynchronously on this stream
                for t in (depth_imgs, rgb_imgs, poses, K) + (() if lazy is not None else (clip_feat_img,)) + tuple(label_maps or ()):
                    t.record_stream(stream)

    def flush(self):
        """Bring the registered buffers up to date: fuse the frames queued behind integrate() and finish a deferred
        clear (reset).  Readers of the volume never need to call it -- every access to a registered buffer does -- it
        exists for callers that hold raw pointers or tensors obtained earlier."""
        self._sync_volume()

    def _sync_volume(self):
        # the frames flushed HERE are the last before somebody looks: the queue's session is finished behind them, the deferred
        # clear of reset() -- still owed after the flushes the queue started on its own: a scan of several windows would otherwise
        # zero, after its first window, most of a volume that the later windows write -- is paid, and the current stream waits
        self._flush_pending(final=True)
        if self.__dict__.get("_feat_stale"):
            # reset() did not clear the feature rows: zero the ones that are still unwritten (weight 0)
            self.__dict__["_feat_stale"] = False
            vol = self._c_volume(for_fuse=True)
            dev = self._buffers["tsdf"].device
            with torch.cuda.device(dev):
                check(lib().saf_clear_unwritten_rows(C.byref(vol), 0, self._buffers["tsdf"].numel(), current_stream_ptr()),
                      "saf_clear_unwritten_rows")

    def _queue_busy(self):
        """Frames staged but not handed over, or a pushed window whose row kernel is still owed."""
        return bool(self.__dict__.get("_pending_n", 0) or self.__dict__.get("_session_open"))

    def _wait_for_queue(self):
        """The current stream waits for what the queue's stream has been given (readers of the volume)."""
        ev = self.__dict__.get("_fs_event")
        if ev is not None:
            dev = self._buffers["tsdf"].device
            torch.cuda.current_stream(dev).wait_event(ev)

    def _finish_session(self):
        """Launch the row kernel of the window the session still holds (on the queue's stream) and let the current stream wait."""
        if not self.__dict__.get("_session_open"):
            self._wait_for_queue()
            return
        dev = self._buffers["tsdf"].device
        fs = self.__dict__["_fs"]
        with torch.cuda.device(dev):
            rc = lib().saf_fuse_session_finish(self.__dict__["_session"], fs.cuda_stream)
            self.__dict__["_session_open"] = False
            ev = fs.record_event()
            self.__dict__["_fs_event"] = ev
            st = self.__dict__.get("_stage")
            if st is not None:  # the open window's frames (and, conservatively, everybody's) are free behind `ev`
                st["free_ev"] = [ev] * len(st["free_ev"])
                st["sess_frames"] = 0
                st["chunk_ready"] = {}
            if rc != 0:  # its windows are classified and partly fused: nothing to retry
                self.__dict__["_poisoned"] = "the queue's session could not launch its last row kernel; the volume is incomplete: reset() it"
            check(rc, "saf_fuse_session_finish")
            torch.cuda.current_stream(dev).wait_event(ev)

    def _prepare_chunk(self, st, lo, n, stream):
        """``saf_fuse_session_prepare`` for the staged frames in ring slots [lo, lo + n): their depth tiles, on ``stream`` behind
        their staging, and the event a later push looks at (``hipEventQuery``).  A shape no session takes: nothing to do."""
        if self.__dict__.get("_poisoned") or getattr(self, "_shard_stripes", None) is not None:
            return
        sl = slice(lo, lo + n)
        labs = None if st["labels"] is None else st["labels"][sl]
        arr, _keep, npy, npx = self._make_frames(st["depth"][sl], st["rgb"][sl], st["pose"][sl], st["K"][sl], st["feat"][sl], labs, st["key"][4],
                                                 borrowed=st["bptr"][sl])
        vol = self._c_volume(for_fuse=True)
        ws = self._get_workspace(npy, npx, (int(st["depth"].shape[1]), int(st["depth"].shape[2])))
        L = lib()
        if L.saf_fuse_session_ok(C.byref(vol), arr, n, ws.numel()) != 1:
            return
        if self.__dict__.get("_session") is None:
            self.__dict__["_session"] = L.saf_fuse_session_create()
        if L.saf_fuse_session_prepare(self.__dict__["_session"], C.byref(vol), arr, n, ws.data_ptr(), ws.numel(), stream.cuda_stream) == 0:
            st.setdefault("chunk_ready", {})[lo + n] = stream.record_event()  # keyed by the slot behind the chunk

    def _flush_pending(self, final=False, keep=0):
        """Hand the staged frames to the library.  ``final``: somebody is about to look -- the session is finished behind them
        (its last window's rows launched, the current stream waits); else the frames but the newest ``keep`` are pushed into
        the session (the newest chunk's staging and depth tiles are still running: it follows one chunk later, wait-free)."""
        n = self.__dict__.get("_pending_n", 0) - (0 if final else keep)
        keep = self.__dict__.get("_pending_n", 0) - n
        if n <= 0:
            if final:
                self._finish_session()
            return
        self._check_poisoned()
        self.__dict__["_pending_n"] = 0  # first: the buffer accesses below must not re-enter
        self.__dict__["_fuse_launched"] = False
        st = self.__dict__["_stage"]
        lo = st["ring0"]
        sl = slice(lo, lo + n)
        try:
            labs = None if st["labels"] is None else st["labels"][sl]
            feat = st["feat"][sl]
            if st.get("lazy") is not None:  # the queued frames' feature maps, in one backbone batch
                fn, ac_on, ac_dtype = st["lazy"]
                with torch.no_grad(), torch.autocast("cuda", dtype=ac_dtype, enabled=ac_on):
                    feat = fn(st["rgb"][sl])
                if tuple(feat.shape) != (n,) + tuple(st["key"][2]):
                    raise SafError(f"the backbone returned {tuple(feat.shape)} for {n} frames, expected {(n,) + tuple(st['key'][2])}")
                # into the ring: the last window's row kernel reads its maps when the NEXT push launches it -- a temporary of
                # this call would be back in the allocator's hands by then (the ring's quarters are guarded by events)
                st["feat"][sl].copy_(feat)
                feat = st["feat"][sl]
            self._fuse_now(st["depth"][sl], st["rgb"][sl], st["pose"][sl], st["K"][sl], feat, labs, st["key"][4], staged=(lo, n, final))
        except BaseException as exc:
            if not self.__dict__.get("_fuse_launched"):
                # the backbone, the descriptors or the argument checks at the entry of the fuse call failed (out of
                # memory, an unsupported tiling) BEFORE any kernel touched the volume: the frames stay queued -- the next
                # access raises again instead of reading a volume that silently lacks them
                self.__dict__["_pending_n"] = n + keep
            else:
                # the call failed after launching some of its windows: fusing the same frames again would count them
                # twice, dropping them would lose them silently.  Neither: the volume is unusable until reset().
                self.__dict__["_poisoned"] = f"a flush of {n} queued frames failed part-way ({exc!r}); the volume is incomplete: reset() it"
            raise
        self.__dict__["_pending_n"] = keep
        if final:
            self._finish_session()
            st["ring0"] = 0  # (every quarter now carries the event of the finish, or of the call on the current stream)
            if any(r is not None for r in st["brefs"]):  # lent images: read by now as far as this stream can tell -- let them go
                st["brefs"] = [None] * len(st["brefs"])
                st["bptr"][:] = 0
        else:
            st["ring0"] = (lo + n) % self._QUEUE_FRAMES  # the next frames follow in the ring (a session's windows are its quarters)

    def _check_poisoned(self):
        msg = self.__dict__.get("_poisoned")
        if msg:
            raise SafError(msg)

    @property
    def pending_frames(self):
        """Frames staged behind integrate() and not yet handed to the fuse call (a completed window is handed over at once)."""
        return self.__dict__.get("_pending_n", 0)

    def __getattr__(self, name):
        # registered buffers live in _buffers, so every read of one comes through here
        if name in _VOLUME_BUFFERS:
            if self.__dict__.get("_poisoned"):
                self._check_poisoned()
            if name == "clip_feat" and self.__dict__.get("_feat_stale"):
                self._sync_volume()  # queued frames AND the deferred clear: the rows are about to be looked at
            elif self.__dict__.get("_pending_n", 0) or self.__dict__.get("_session_open"):
                self._flush_pending(final=True)
            elif self.__dict__.get("_fs_event") is not None:
                self._wait_for_queue()  # (a reader on another stream than the one that finished the session)
        return super().__getattr__(name)

    def __setattr__(self, name, value):
        if (name in _VOLUME_BUFFERS or name == "accum_mode") and (self._queue_busy() or self.__dict__.get("_feat_stale")):
            self._sync_volume()  # queued frames belong to the buffers / mode that were current when they were queued
        super().__setattr__(name, value)

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .cpu() / .float()
        if self._queue_busy() or self.__dict__.get("_feat_stale"):
            self._sync_volume()
        self._wait_for_queue()
        self.__dict__["_stage"] = None
        self.__dict__["_fs"] = None  # (a stream of the old device)
        self.__dict__["_fs_event"] = None
        self.__dict__["_fs_seen"] = None
        return super()._apply(fn, *args, **kwargs)

    def _save_to_state_dict(self, *args, **kwargs):
        self._sync_volume()
        return super()._save_to_state_dict(*args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._sync_volume()
        out = super()._load_from_state_dict(*args, **kwargs)
        # The windowed path never reads the feature row of a voxel whose weight is 0 (such a row is zero by construction).  A
        # loaded state need not keep that promise: the rows of weight-0 voxels are zeroed at the next access (the deferred
        # clear of reset(): saf_clear_unwritten_rows), so the invariant is enforced, not assumed.
        self.__dict__["_feat_stale"] = True
        return out

    def named_buffers(self, *args, **kwargs):
        self._sync_volume()
        return super().named_buffers(*args, **kwargs)

    def _queue_stream(self, dev, tensors):
        """The queue's own stream (created at the first push); ``tensors`` it is about to use are marked for the allocator
        (record_stream, once per allocation: whoever frees them -- a deleted module, a regrown workspace -- must not hand the
        memory out while that stream still reads it)."""
        fs = self.__dict__.get("_fs")
        if fs is None:
            fs = torch.cuda.Stream(device=dev)
            self.__dict__["_fs"] = fs
            self.__dict__["_fs_seen"] = set()
        seen = self.__dict__.get("_fs_seen")
        if seen is None:
            seen = self.__dict__["_fs_seen"] = set()
        for t in tensors:
            if t is not None and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                t.record_stream(fs)
        return fs

    def _fuse_now(self, depth_imgs, rgb_imgs, poses, K, clip_feat_img, label_maps=None, rgb_bilinear=False, staged=None):
        """``staged`` = (first ring slot, frames, final): the frames lie in the queue's staging ring -- where the windowed
        two-stream path takes them they are PUSHED into the queue's session on its own stream (their last window's row kernel
        stays owed: ``_finish_session``); everything else is one call on the current stream, behind the session."""
        if getattr(self, "_shard_stripes", None) is not None:
            raise SafError(
                "this volume holds only its reduce-scattered voxel stripes "
                f"({len(self._shard_stripes)} of them) of a merged job; all_gather it (distributed.gather_shards, or "
                "merge_volumes(..., gather=True)) before fusing more frames"
            )
        arr, keep, npy, npx = self._make_frames(depth_imgs, rgb_imgs, poses, K, clip_feat_img, label_maps, rgb_bilinear,
                                                borrowed=None if staged is None else self.__dict__["_stage"]["bptr"][staged[0]:staged[0] + staged[1]])
        vol = self._c_volume(for_fuse=True)
        ws = self._get_workspace(npy, npx, (int(depth_imgs.shape[1]), int(depth_imgs.shape[2])))
        L = lib()
        windowed = L.saf_fuse_path(C.byref(vol), arr, len(arr), ws.numel()) == 1
        # the module's device, not the caller's current one, owns the launch (and its current stream)
        dev = self._buffers["tsdf"].device
        st = self.__dict__.get("_stage")
        win = _abi.SAF_WINDOW_FRAMES
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev)
            # (a flush of a few frames with no window open takes the per-frame pipeline, as a direct call of that size does)
            if staged is not None and (self.__dict__.get("_session_open") or len(arr) >= 16) and \
                    L.saf_fuse_session_ok(C.byref(vol), arr, len(arr), ws.numel()) == 1:
                lo, n, _final = staged
                b = self._buffers
                fs = self._queue_stream(dev, [ws, st["depth"], st["rgb"], st["pose"], st["K"], st["feat"], st["labels"]] +
                                        [b.get(k) for k in ("tsdf", "tsdf_weight", "weight", "rgb", "clip_feat", "labels_one_hot",
                                                            "axis_x", "axis_y", "axis_z", "fuse_stats")])
                # the staging kernels (and a deferred backbone's maps) of these frames: what their classification waits for -- NOT
                # the queue's stream, where the previous window's row kernel sits (the launches are to run beside it).  The
                # session's first push also orders the queue's stream behind the caller's (reset()'s zeroing, an earlier finish)
                # (a chunk whose tiles were prepared a chunk ago brings the event recorded behind them -- completed by now, the
                #  library finds, and waits for nothing; anything else: an event behind what this stream holds now)
                ready = (st.get("chunk_ready") or {}).pop(lo + n, None)
                if ready is None or st.get("lazy") is not None:  # (a deferred backbone has just written these frames' maps: behind THAT)
                    ready = stream.record_event()
                for k in [k for k in (st.get("chunk_ready") or {}) if k <= lo + n]:
                    del st["chunk_ready"][k]
                if not self.__dict__.get("_session_open"):
                    fs.wait_event(ready)
                if self.__dict__.get("_session") is None:
                    self.__dict__["_session"] = L.saf_fuse_session_create()
                # (the depth tiles of these frames: on the caller's stream, behind their staging -- out of the classification chain)
                rc = L.saf_fuse_session_push(self.__dict__["_session"], C.byref(vol), arr, len(arr), ws.data_ptr(), ws.numel(),
                                             b["fuse_stats"].data_ptr(), fs.cuda_stream, ready.cuda_event, stream.cuda_stream)
                st.setdefault("ready_evs", []).append(ready)  # (kept alive while the device may still wait for them)
                del st["ready_evs"][:-8]
                # SAF_E_INVALID / _WORKSPACE / _UNSUPPORTED come from the checks at the entry (every frame descriptor is validated
                # before the first launch); a HIP error may have left some windows fused (see _flush_pending)
                self.__dict__["_fuse_launched"] = rc == 0 or rc == _abi.SAF_E_HIP
                if rc == 0:
                    self.__dict__["_session_open"] = True
                    ev = fs.record_event()
                    self.__dict__["_fs_event"] = ev
                    # the session's windows are the ring's quarters (a session starts at slot 0): the windows this push completed
                    # have their row kernels queued -- behind `ev` their frames are free
                    w0, w1 = st["sess_frames"] // win, (st["sess_frames"] + n) // win
                    for w in range(w0, w1):
                        st["free_ev"][w % len(st["free_ev"])] = ev
                    st["sess_frames"] += n
                check(rc, "saf_fuse_session_push")
                return
            # one call on the current stream: whatever the session still owes comes first
            self._finish_session()
            # after a lazy reset() the feature rows still hold the previous scan: the windowed path never reads a weight-0 row
            # (the clear stays owed until somebody looks: _sync_volume); the per-frame pipeline reads the rows it updates --
            # saf_fuse_frames_recycled zeroes the unwritten rows first
            stale = bool(self.__dict__.get("_feat_stale")) and not windowed
            if stale:
                rc = L.saf_fuse_frames_recycled(
                    C.byref(vol), arr, len(arr), ws.data_ptr(), ws.numel(), self._buffers["fuse_stats"].data_ptr(), None, stream.cuda_stream
                )
            else:
                rc = L.saf_fuse_frames(
                    C.byref(vol), arr, len(arr), ws.data_ptr(), ws.numel(), self._buffers["fuse_stats"].data_ptr(), stream.cuda_stream
                )
            if stale and rc == 0:  # (the recycled call: every weight-0 row is zero behind it)
                self.__dict__["_feat_stale"] = False
            self.__dict__["_fuse_launched"] = rc == 0 or rc == _abi.SAF_E_HIP
            check(rc, "saf_fuse_frames")
            if staged is not None:  # the ring's slots of these frames are read on this stream
                ev = stream.record_event()
                for q in range(staged[0] // win, (staged[0] + staged[1] - 1) // win + 1):
                    st["free_ev"][q] = ev
            # the launches are asynchronous: keep inputs alive until the stream has consumed them
            for t in keep[:5]:
                t.record_stream(stream)
            for t in keep[5] or ():
                t.record_stream(stream)

    # -- extensions (not in the reference) ---------------------------------------------------
    def reset(self, accum_mode=_abi.SAF_RUNNING_MEAN, lazy=True):
        """Back to the freshly constructed state: every volume buffer zero (the reference builds a new module per
        scan, clip_seem_fusion.py:291-302).  ``lazy``: the 4*D*N bytes of ``clip_feat`` are not cleared here -- a voxel
        with weight 0 has a zero row by contract and the windowed fuse path never reads such rows; the rows still
        unwritten are zeroed when something first looks (any buffer access, state_dict, the merge, flush())."""
        self.__dict__["_pending_n"] = 0  # frames still queued would be fused into a volume that is being discarded
        self.__dict__["_poisoned"] = None
        self.__dict__["_feat_stale"] = False
        b = self._buffers
        fs = self.__dict__.get("_fs")
        if fs is not None:
            # a pushed window whose row kernel is still owed is dropped with the volume; what the queue's stream (and the
            # library's classification stream behind it) already runs must finish before the buffers are zeroed here
            if self.__dict__.get("_session") is not None:
                check(lib().saf_fuse_session_abandon(self.__dict__["_session"]), "saf_fuse_session_abandon")
            self.__dict__["_session_open"] = False
            ev = fs.record_event()
            self.__dict__["_fs_event"] = ev
            torch.cuda.current_stream(b["tsdf"].device).wait_event(ev)
            st = self.__dict__.get("_stage")
            if st is not None:
                st["sess_frames"], st["ring0"] = 0, 0
                st["free_ev"] = [ev] * len(st["free_ev"])
                st["chunk_ready"] = {}
        lazy = bool(lazy) and b["clip_feat"].is_cuda
        for name in ("rgb", "tsdf", "weight", "tsdf_weight", "labels_one_hot") + (() if lazy else ("clip_feat",)):
            t = b.get(name)
            if t is not None:
                t.zero_()
        super().__setattr__("accum_mode", accum_mode)
        self._shard_stripes = None
        self.__dict__["_shard_plans"] = None
        self.__dict__["_feat_stale"] = lazy


    def integrate_features(self, depth_imgs, rgb_imgs, poses, K, clip_feat_img, label_maps=None):
        """``integrate`` with the backbone outputs supplied by the caller (fuse-only entry used by
        the benchmark and the parity tests)."""
        self._fuse(depth_imgs, rgb_imgs, poses, K, clip_feat_img, label_maps, self._rgb_bilinear)

    def stats(self):
        """dict of counters accumulated by the kernels (forces a device sync)."""
        s = self.fuse_stats.cpu().tolist()
        if s[4]:
            raise SafError(f"{s[4]} fuse workgroups timed out waiting for their frame's sweep; the volume is incomplete")
        # the form the windowed path takes for this volume NOW (environment, read per call by the library): results of two
        # runs are comparable bit for bit only under the same form
        form = os.environ.get("SAF_WIN_FORM", "sums")
        form = {"s": "sums", "r": "rows", "b": "bricks"}.get(form[:1], "sums")
        if form == "sums" and self._buffers["clip_feat"].dtype == torch.bfloat16:
            form = "rows (SAF_WIN_MAPS16=0)" if os.environ.get("SAF_WIN_MAPS16", "1")[:1] == "0" else "sums, bf16 map images"
        # the windowed path's frame cull: (brick, frame) pairs tested / dropped by reason (include/saf.h, stats[8..12])
        cull = {"pairs": s[8], "behind": s[9], "far": s[10], "frustum": s[11], "occluded": s[12]}
        return {"valid": s[0], "tsdf_valid": s[1], "frames": s[2], "labels_dropped": s[3], "window_rows": s[5],
                "window_tsdf_voxels": s[6], "window_form": form, "cull": cull}

    def sample_mesh_vertices(self, verts_index, voxel_obj_idx=None, objects_segmentation_color=None):
        """The sampling half of ``extract_mesh`` (reference clipfusion.py:741-760, clip_seem_fusion.py:843-878)
        on the HIP device: marching-cubes vertices in voxel-index coordinates -> (vertex_colors[V,3] clamped,
        vertex_clip_feats[V,D] f32[, vertex_obj_idx[V,1], vertex_segment_color[V,3]])."""
        verts = torch.as_tensor(np.asarray(verts_index, dtype=np.float32)).to(self.tsdf.device).contiguous()
        nv = verts.shape[0]
        dev = self.tsdf.device
        feat = torch.empty((nv, int(self.n_clip_feats)), dtype=torch.float32, device=dev)
        rgb = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        oi = so = oo = sc = None
        if voxel_obj_idx is not None:
            oi = torch.as_tensor(voxel_obj_idx).to(device=dev, dtype=torch.int32).contiguous().reshape(-1)
            oo = torch.empty(nv, dtype=torch.float32, device=dev)
        if objects_segmentation_color is not None:
            sc = self._f32c(torch.as_tensor(objects_segmentation_color).to(dev), "objects_segmentation_color")
            so = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        vol = self._c_volume()
        p = _abi.ptr
        with torch.cuda.device(dev):
            rc = lib().saf_sample_vertices(C.byref(vol), p(verts), nv, p(feat), p(rgb), p(oi), p(oo), p(sc), p(so),
                                           current_stream_ptr())
        check(rc, "saf_sample_vertices")
        out = [rgb, feat]
        if oo is not None:
            out.append(oo[:, None])
        if so is not None:
            out.append(so)
        return tuple(out)

    def _marching_cubes_vertices(self, marching_cubes=None):
        """TSDF -> (verts in index space, faces) (clipfusion.py:724-739): un-fused voxels act as NaN, faces touching
        a NaN vertex are dropped, vertices re-indexed.  Default: on the device (saf_marching_cubes_*), nothing but the
        mesh leaves it.  ``marching_cubes=`` takes a callable with scikit-image's signature (e.g.
        ``skimage.measure.marching_cubes``) to run the reference's own CPU route instead."""
        if marching_cubes is None:
            nx, ny, nz = (int(v) for v in self.nvox)
            verts, faces = marching_cubes_gpu(self.tsdf.view(nx, ny, nz), self.weight.view(nx, ny, nz), level=0.0)
            return verts.cpu().numpy(), faces.cpu().numpy().astype(np.int64)
        tsdf = self.tsdf.masked_fill(self.weight == 0, torch.nan).cpu()
        verts, faces = marching_cubes(tsdf.view(*[int(v) for v in self.nvox]).numpy(), level=0)[:2]
        faces = faces[~np.isnan(verts[faces]).any(axis=(1, 2))]
        used = np.zeros(len(verts), dtype=bool)
        used[np.unique(faces)] = True
        faces = (np.cumsum(used) - 1)[faces]
        return verts[used], faces

    def _verts_world(self, verts):
        return verts * self.voxel_size + torch.as_tensor(self.origin).cpu().numpy()

    def extract_mesh(self, marching_cubes=None):
        """Reference clipfusion.py:723-763: (verts_world, faces, vertex_colors, vertex_clip_feats).  Marching cubes
        and the vertex sampling of the D-channel volume both run in HIP; the volume never leaves the device."""
        verts, faces = self._marching_cubes_vertices(marching_cubes)
        colors, feats = self.sample_mesh_vertices(verts)[:2]
        return self._verts_world(verts), faces, colors, feats


# --------------------------------------------------------------------------------------------
# Clip: tiled image features (PyTorch-ROCm backbone) + the text query head (HIP scan)
# --------------------------------------------------------------------------------------------


def _norm_mode(normalize):
    """False/0 -> none; True/1 -> L2 + nan_to_num (clip_seem_fusion.py:507-511); 2 / "clamp" -> divide by
    max(norm, 0.1) as eval_scannet_segmentation.py:549-551 and hypersim_eval.py:50-51 do."""
    if normalize in ("clamp", "clamp_min"):
        return _abi.SAF_NORM_L2_CLAMP
    if isinstance(normalize, bool):
        return int(normalize)
    mode = int(normalize)
    if mode not in (_abi.SAF_NORM_NONE, _abi.SAF_NORM_L2, _abi.SAF_NORM_L2_CLAMP):
        raise ValueError(f"bad normalize mode {normalize!r}")
    return mode


def _query_scan(feats, text, epilogue, scale=1.0, normalize=False, last_only=False):
    """Run saf_query_scan on [N,D] f32 features and [L,>=D] f32 text embeddings (both moved to the
    HIP device if needed); returns [N,L] (or [N] when last_only) on the device of ``feats``."""
    if not torch.cuda.is_available():
        raise SafError("the query scan needs the MI355X device; there is no CPU fallback")
    out_dev = feats.device
    dev = feats.device if feats.is_cuda else torch.device("cuda", torch.cuda.current_device())
    f = feats.detach().to(device=dev)
    ft = {torch.float32: _abi.SAF_F32, torch.bfloat16: _abi.SAF_BF16, torch.float16: _abi.SAF_F16}.get(f.dtype)
    if ft is None:
        f, ft = f.float(), _abi.SAF_F32
    if f.dim() != 2:
        raise ValueError("features must be [N,D]")
    if f.stride(1) != 1:
        f = f.contiguous()
    t = text.detach().to(device=dev, dtype=torch.float32)
    if t.stride(1) != 1:
        t = t.contiguous()
    n, d = f.shape
    nl = t.shape[0]
    if t.shape[1] < d:
        raise RuntimeError(f"text features have {t.shape[1]} dims, image features {d}")
    out = None if last_only else torch.empty((n, nl), dtype=torch.float32, device=dev)
    last = torch.empty(n, dtype=torch.float32, device=dev) if last_only else None
    wsb = lib().saf_query_workspace_bytes(nl, epilogue)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
    with torch.cuda.device(dev):
        rc = lib().saf_query_scan(
            f.data_ptr(), ft, n, f.stride(0), d, t.data_ptr(), nl, t.stride(0), epilogue, float(scale),
            _norm_mode(normalize), _abi.ptr(out), _abi.ptr(last), _abi.ptr(ws), wsb, current_stream_ptr(),
        )
    check(rc, "saf_query_scan")
    res = last if last_only else out
    return res.to(out_dev)


_DT = {torch.float32: _abi.SAF_F32, torch.bfloat16: _abi.SAF_BF16, torch.float16: _abi.SAF_F16}


def marching_cubes_gpu(tsdf, weight, level=0.0):
    """Marching cubes on the device (saf_marching_cubes_count / _emit): ``tsdf`` [nx,ny,nz] f32 and ``weight`` [nx,ny,nz]
    i32 on the HIP device -> (verts [V,3] f32 in voxel-index coordinates, faces [F,3] i32), both on the device.
    Voxels with weight 0 are the reference's NaN mask (clipfusion.py:724); faces with a vertex on an edge to such a
    voxel do not exist, nor do unused vertices (clipfusion.py:730-739)."""
    require_cuda(tsdf, "tsdf")
    require_cuda(weight, "weight")
    if tsdf.dim() != 3 or tuple(weight.shape) != tuple(tsdf.shape):
        raise ValueError("tsdf and weight must be [nx,ny,nz]")
    t = tsdf.float().contiguous()
    w = weight.to(torch.int32).contiguous()
    nx, ny, nz = (int(v) for v in t.shape)
    dev = t.device
    L = lib()
    if min(nx, ny, nz) < 2:
        return torch.zeros((0, 3), device=dev), torch.zeros((0, 3), dtype=torch.int32, device=dev)
    wsb = L.saf_marching_cubes_workspace_bytes(nx * ny * nz)
    if wsb == 0:
        raise SafError("marching cubes: the grid is too large (2^31 voxels or more)")
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(L.saf_marching_cubes_count(t.data_ptr(), w.data_ptr(), nx, ny, nz, float(level), ws.data_ptr(), wsb,
                                         counts.data_ptr(), current_stream_ptr()), "saf_marching_cubes_count")
        nv, nf = (int(v) for v in counts.tolist())  # the sizes are results: one read-back
        verts = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        faces = torch.empty((nf, 3), dtype=torch.int32, device=dev)
        check(L.saf_marching_cubes_emit(t.data_ptr(), w.data_ptr(), nx, ny, nz, float(level), ws.data_ptr(), wsb,
                                        _abi.ptr(verts) if nv else None, nv, _abi.ptr(faces) if nf else None, nf,
                                        current_stream_ptr()), "saf_marching_cubes_emit")
    return verts, faces


_WIDE_EPILOGUES = {"scores": _abi.SAF_QW_SCORES, "vs_background": _abi.SAF_QW_VS_BACKGROUND,
                   "row_argmax": _abi.SAF_QW_ROW_ARGMAX, "query_max": _abi.SAF_QW_QUERY_MAX}


def query_scan_wide(feats, text, epilogue="scores", scale=1.0, normalize=True, n_background=0, rescale=False,
                    out_dtype=None, row_offset=0, out=None):
    """Many text queries over a 16-bit feature volume on the 16-bit matrix cores (BASELINE config 5: the
    query_mesh.py / hypersim_eval.py scan with ~1000 targets), with the callers' reductions fused into the scan
    (saf_query_scan_wide_ex) so that the N x Q score matrix -- twice the size of the volume -- need not exist.

    ``feats`` [N, D] float16 / bfloat16 on the HIP device, D in {256, 512} (128: scores only); ``text`` [Q, >=D] fp32
    (rounded to the feature dtype).  ``epilogue``:

    * ``"scores"``         -> [N, Q]: ``scale * <f_n / |f_n|, t_q>``
    * ``"vs_background"``  -> [N, Q - n_background]: ``softmax(scale * [<f, bg_0..>, <f, target>])[-1]`` for every target,
      the first ``n_background`` text rows being the shared background prompts (query_mesh.py:36-39 with
      ``scale=100``; hypersim_eval.py:76-81); ``rescale=True`` adds ``((r - 0.5) * 2).clamp(0, 1)`` (query_mesh.py:39)
    * ``"row_argmax"``     -> (index int32 [N], value f32 [N]): best query per row
      (eval_scannet_segmentation.py:553-560, the first label of the argsort)
    * ``"query_max"``      -> (value f32 [Q], row int64 [Q]): best row per query, rows numbered from ``row_offset``

    ``out``: an optional preallocated [N, columns] tensor for the two matrix-valued epilogues (row stride a multiple
    of 8 elements keeps the 16-byte stores of the epilogue aligned; a multiple of 128 bytes is 3-4 % faster at many columns).
    """
    require_cuda(feats, "features")
    if feats.dtype not in (torch.float16, torch.bfloat16) or feats.dim() != 2:
        raise ValueError("query_scan_wide needs a [N, D] float16 / bfloat16 feature tensor")
    if feats.stride(1) != 1:
        feats = feats.contiguous()
    epi = _WIDE_EPILOGUES[epilogue] if isinstance(epilogue, str) else int(epilogue)
    out_dtype = out_dtype or feats.dtype
    dev = feats.device
    t = text.detach().to(device=dev, dtype=torch.float32)
    if t.stride(1) != 1:
        t = t.contiguous()
    n, d = feats.shape
    q = t.shape[0]
    if t.shape[1] < d:
        raise RuntimeError(f"text features have {t.shape[1]} dims, image features {d}")
    L = lib()
    norm = _norm_mode(normalize)
    if d == 128 and epi == _abi.SAF_QW_SCORES:  # the narrow shape keeps the first kernel
        out = torch.empty((n, q), dtype=out_dtype, device=dev)
        wsb = L.saf_query_wide_workspace_bytes(q, d)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.saf_query_scan_wide(feats.data_ptr(), _DT[feats.dtype], n, feats.stride(0), d, t.data_ptr(), q,
                                       t.stride(0), float(scale), norm, out.data_ptr(), _DT[out_dtype], out.stride(0),
                                       ws.data_ptr(), wsb, current_stream_ptr())
        check(rc, "saf_query_scan_wide")
        return out
    idx = val = row = None
    if epi in (_abi.SAF_QW_SCORES, _abi.SAF_QW_VS_BACKGROUND):
        cols = q - (int(n_background) if epi == _abi.SAF_QW_VS_BACKGROUND else 0)
        if out is None:
            # rows padded to a multiple of 8 columns keep every 16-byte store of the epilogue aligned; wide outputs are padded to
            # whole 128-byte lines (a tile's 64-byte segment then never straddles two lines: 3-4 % of the scan at 1000 columns,
            # tools/probe_out_stride.py).  The result is the [:, :cols] view
            esz = torch.empty((), dtype=out_dtype).element_size()
            per_line = 128 // esz
            stride = (cols + per_line - 1) // per_line * per_line if cols >= 4 * per_line else (cols + 7) // 8 * 8
            out = torch.empty((n, stride), dtype=out_dtype, device=dev)[:, :cols]
        else:
            require_cuda(out, "out")
            if tuple(out.shape) != (n, cols) or out.stride(1) != 1:
                raise ValueError(f"out must be [{n}, {cols}] with unit column stride")
            out_dtype = out.dtype
    elif epi == _abi.SAF_QW_ROW_ARGMAX:
        out = None
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        val = torch.empty(n, dtype=torch.float32, device=dev)
    else:
        out = None
        val = torch.empty(q, dtype=torch.float32, device=dev)
        row = torch.empty(q, dtype=torch.int64, device=dev)
    wsb = L.saf_query_wide_ex_workspace_bytes(q, d, epi, int(n_background))
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.saf_query_scan_wide_ex(
            feats.data_ptr(), _DT[feats.dtype], n, feats.stride(0), d, t.data_ptr(), q, t.stride(0), float(scale), norm,
            epi, int(n_background), int(bool(rescale)), _abi.ptr(out), _DT[out_dtype], 0 if out is None else out.stride(0),
            _abi.ptr(idx), _abi.ptr(val), _abi.ptr(row), int(row_offset), ws.data_ptr(), ws.numel(), current_stream_ptr(),
        )
    check(rc, "saf_query_scan_wide_ex")
    if out is not None:
        return out
    return (idx, val) if idx is not None else (val, row)


def query_scores_wide(feats, text, scale=1.0, normalize=True, out_dtype=None):
    """Cosine scores of many text queries over a 16-bit feature volume: ``out[n, q] = scale * <f_n / |f_n|, t_q>``
    ([N, Q] of ``out_dtype``, default feats.dtype).  See query_scan_wide for the fused reductions."""
    return query_scan_wide(feats, text, "scores", scale=scale, normalize=normalize, out_dtype=out_dtype)


class Clip(torch.nn.Module):
    """Mirror of the reference ``Clip`` (clipfusion.py:766-1039).

    ``backbone``/``tokenizer`` may be injected (any object with ``encode_image``, ``encode_text``
    and ``visual.output_dim``); by default they come from ``open_clip`` exactly as in the
    reference (clipfusion.py:769-772).  The ViT GEMMs stay PyTorch-ROCm (hipBLASLt / MFMA).
    """

    # tiles per encode_image call.  The reference caps at 8 (clipfusion.py:826, sized for a 24 GB card); with 288 GB of HBM
    # the ViT GEMMs are fed 1024 tiles at a time: 468 -> 870-1000 frames/s end to end with a bf16 ViT-B/32 (bench.py --end-to-end)
    max_patch_batch_size = 1024

    def __init__(self, clip_model, pretraining, backbone=None, tokenizer=None):
        super().__init__()
        if backbone is None:
            try:
                import open_clip
            except ImportError as e:  # same failure mode as the reference's module import
                raise ImportError(
                    "open_clip is required to build the CLIP backbone (or pass backbone=/tokenizer=)"
                ) from e
            backbone = open_clip.create_model(clip_model, pretrained=pretraining, require_pretrained=True)
            tokenizer = open_clip.get_tokenizer(clip_model)
        self.clip = backbone
        self.tokenizer = tokenizer
        self.channel_mean = torch.nn.Parameter(
            torch.tensor([0.48145466, 0.4578275, 0.40821073])[None, :, None, None], requires_grad=False
        )
        self.channel_std = torch.nn.Parameter(
            torch.tensor([0.26862954, 0.26130258, 0.27577711])[None, :, None, None], requires_grad=False
        )
        self.feature_dim = self.clip.visual.output_dim

    def normalize_img(self, rgb_img_0_1):
        return (rgb_img_0_1 - self.channel_mean) / self.channel_std

    def unnormalize_img(self, rgb_img_normed):
        return rgb_img_normed * self.channel_std + self.channel_mean

    def get_patches(self, rgb_imgs, patch_size, patch_stride):
        """[B,3,H,W] -> [B,npy,npx,3,p,p] overlapping tiles (clipfusion.py:789-806)."""
        _, _, imheight, imwidth = rgb_imgs.shape
        assert (imheight - patch_size) % patch_stride == 0
        assert (imwidth - patch_size) % patch_stride == 0
        tiles = rgb_imgs.unfold(2, patch_size, patch_stride).unfold(3, patch_size, patch_stride)
        return tiles.permute(0, 2, 3, 1, 4, 5)

    def tiles_224(self, rgb_imgs, patch_size, patch_stride, dtype=None):
        """[B,3,H,W] in 0..1 -> the [B*npy*npx, 3, 224, 224] batch the ViT consumes (clipfusion.py:808-823: normalise,
        unfold, resize).  On the HIP device one fused kernel (saf_clip_tiles) writes it once, in ``dtype`` (default:
        the autocast dtype when autocast is on, else fp32); on the CPU the reference's three PyTorch steps."""
        _, _, imheight, imwidth = rgb_imgs.shape
        assert (imheight - patch_size) % patch_stride == 0
        assert (imwidth - patch_size) % patch_stride == 0
        if not rgb_imgs.is_cuda:
            patches = self.get_patches(self.normalize_img(rgb_imgs), patch_size, patch_stride)
            bsz, npy, npx = patches.shape[:3]
            patches = patches.reshape(bsz * npy * npx, 3, patch_size, patch_size)
            return torch.nn.functional.interpolate(patches, size=(224, 224), mode="bilinear", align_corners=False)
        if dtype is None:
            dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
        x = rgb_imgs if rgb_imgs.dtype == torch.float32 else rgb_imgs.float()
        bsz = x.shape[0]
        npy, npx = 1 + (imheight - patch_size) // patch_stride, 1 + (imwidth - patch_size) // patch_stride
        out = torch.empty((bsz * npy * npx, 3, 224, 224), dtype=dtype, device=x.device)
        mean = (C.c_float * 3)(*self.channel_mean.detach().reshape(-1).tolist())
        std = (C.c_float * 3)(*self.channel_std.detach().reshape(-1).tolist())
        sb, sc, sy, sx = x.stride()
        with torch.cuda.device(x.device):
            rc = lib().saf_clip_tiles(x.data_ptr(), bsz, imheight, imwidth, sb, sc, sy, sx, int(patch_size), int(patch_stride),
                                      224, mean, std, out.data_ptr(), _DT[dtype], current_stream_ptr())
        check(rc, "saf_clip_tiles")
        return out

    def img_inference_tiled(self, rgb_imgs, patch_size, patch_stride):
        """[B,3,H,W] in 0..1 -> [B,D,npy,npx] CLIP embedding per tile (clipfusion.py:808-839)."""
        bsz = rgb_imgs.shape[0]
        npy = 1 + (rgb_imgs.shape[2] - patch_size) // patch_stride
        npx = 1 + (rgb_imgs.shape[3] - patch_size) // patch_stride
        per_frame = npy * npx
        feats = torch.empty(bsz * per_frame, self.feature_dim, device=rgb_imgs.device)
        step = int(self.max_patch_batch_size)
        # The tile batch is produced in slices of whole frames -- at most max_patch_batch_size tiles (and never more than the
        # tile kernel's 65535 per launch) exist at a time: the deferred-backbone queue hands over up to 512 frames at once,
        # whose tiles would be tens of GB (a fine tiling has hundreds of tiles per frame).
        fpb = max(1, min(step, 65535) // per_frame)
        for f0 in range(0, bsz, fpb):
            patches = self.tiles_224(rgb_imgs[f0 : f0 + fpb], patch_size, patch_stride)
            base = f0 * per_frame
            for start in range(0, len(patches), step):
                cur = patches[start : start + step]
                feats[base + start : base + start + len(cur)] = self.clip.encode_image(cur)
        return feats.view(bsz, npy, npx, self.feature_dim).permute(0, 3, 1, 2)

    def img_inference_tiled_depthscaled(self, rgb_imgs, depth_imgs, K, patch_stride, footprint_m=0.5):
        """[B,3,H,W] in 0..1, depth [B,H,W] m, K [B,3,3] -> [B,D,H,W]: a FULL-resolution feature image (clipfusion.py:841-890;
        ``scale_patches_by_depth``, off everywhere in the reference: :1097, clip_seem_fusion.py:167).  One tile per lattice
        point (every ``patch_stride`` pixels, the first row / column excluded) with a valid depth, sized to cover
        ``footprint_m`` metres at that depth (``round(f * 0.5 / depth)`` pixels, halves rounded down on both sides, clipped at
        the image's upper-left border only -- as the reference slices), resized to 224 x 224 and encoded; every pixel gets the
        MEAN of the embeddings of the tiles that cover it, 0 where none does.  All of an image's tiles go through the ViT in
        batches of ``max_patch_batch_size`` (the reference encodes them one by one).  The reference itself only runs for
        B = 1 (its last line divides [B, D, H, W] by [B, H, W]); any B works here."""
        x = self.normalize_img(rgb_imgs)
        bsz, _, h, w = x.shape
        dev = x.device
        ys = torch.arange(patch_stride, h, patch_stride, device=dev)
        xs = torch.arange(patch_stride, w, patch_stride, device=dev)
        out = torch.zeros(bsz, self.feature_dim, h, w, device=dev)
        step = int(self.max_patch_batch_size)
        for b in range(bsz):
            d = depth_imgs[b][ys][:, xs]                      # depth at the lattice points [ny, nx]
            keep = (d > 0).nonzero()
            if keep.numel() == 0:
                continue
            dk = d[keep[:, 0], keep[:, 1]]
            half_h = ((K[b, 1, 1] * footprint_m / dk).round().to(torch.int64) // 2).tolist()
            half_w = ((K[b, 0, 0] * footprint_m / dk).round().to(torch.int64) // 2).tolist()
            yc, xc = ys[keep[:, 0]].tolist(), xs[keep[:, 1]].tolist()
            boxes = [(max(0, y - hh), min(h, y + hh), max(0, x_ - hw), min(w, x_ + hw))
                     for y, x_, hh, hw in zip(yc, xc, half_h, half_w)]
            if any(y1 <= y0 or x1 <= x0 for y0, y1, x0, x1 in boxes):
                raise ValueError("a depth-scaled tile is empty (depth too large for this focal length); the reference's "
                                 "interpolate call raises for it as well")
            tiles = torch.cat([F.interpolate(x[b : b + 1, :, y0:y1, x0:x1], size=(224, 224), mode="bilinear", align_corners=False)
                               for y0, y1, x0, x1 in boxes])
            emb = torch.cat([self.clip.encode_image(tiles[s0 : s0 + step]) for s0 in range(0, len(tiles), step)]).to(out.dtype)
            cover = torch.zeros(h, w, device=dev)
            for (y0, y1, x0, x1), e in zip(boxes, emb):     # lattice order: the reference's order of additions
                out[b, :, y0:y1, x0:x1] += e[:, None, None]
                cover[y0:y1, x0:x1] += 1
            out[b] /= cover + (cover == 0)
        return out

    def text_inference(self, str_list):
        device = next(self.clip.parameters()).device
        tokens = self.tokenizer(str_list).to(device)
        feats = self.clip.encode_text(tokens)
        return feats / feats.norm(dim=-1, keepdim=True)

    def run_query(self, img_feats, labels):
        """softmax(100 * F @ T^T) over the labels (clipfusion.py:899-904), fused on the GPU."""
        d = img_feats.shape[-1]
        text = self.text_inference(labels)[:, :d]
        flat = img_feats.reshape(-1, d)
        rel = _query_scan(flat, text, _abi.SAF_Q_SOFTMAX, scale=100.0)
        return rel.reshape(*img_feats.shape[:-1], text.shape[0])

    @staticmethod
    def clip_feature_surgery(image_features, text_features, redundant_feats=None, t=2):
        """[b,n,c] x [t,c] -> [b,n,t] (clipfusion.py:906-934) without the [b,n,t,c] temporary."""
        if image_features.dim() != 3:
            raise ValueError("image_features must be [b,n,c]")
        if redundant_feats is not None:
            text = text_features - redundant_feats.to(text_features.device)
            return torch.stack([_query_scan(f, text, _abi.SAF_Q_SCORES) for f in image_features])
        if image_features.shape[0] != 1:
            # the reference's `w.reshape(1, 1, n_t, 1)` only works for b == 1
            raise RuntimeError("clip_feature_surgery without redundant_feats needs a batch of 1")
        return _query_scan(image_features[0], text_features, _abi.SAF_Q_SURGERY)[None]

    def encode_text_with_prompt_ensemble(self, texts, device, prompt_templates=None):
        """Mean of L2-normalised prompt embeddings per class, renormalised (clipfusion.py:936-1039)."""
        if prompt_templates is None:
            prompt_templates = IMAGENET_PROMPT_TEMPLATES
        model_device = next(self.clip.parameters()).device
        feats = []
        for text in texts:
            tokens = self.tokenizer([tpl.format(text) for tpl in prompt_templates])
            emb = self.clip.encode_text(tokens.to(model_device))
            emb = emb / emb.norm(dim=-1, keepdim=True)
            emb = emb.mean(dim=0)
            feats.append(emb / emb.norm())
        return torch.stack(feats, dim=0).to(device)


def _imagenet_templates():
    """The 85 prompt templates the reference uses by default (OpenAI's ImageNet prompt-engineering
    set plus five '... in the scene.' scene prompts, clipfusion.py:939-1025), generated from their
    regular structure rather than listed."""
    a_the = lambda pat: [pat.format(art) for art in ("a", "the")]
    singles = [
        "a photo of many {}.", "a photo of my {}.", "a photo of one {}.", "itap of my {}.",
        "there is a {} in the scene.", "there is the {} in the scene.", "this is a {} in the scene.",
        "this is the {} in the scene.", "this is one {} in the scene.",
    ]
    paired = [
        "a bad photo of {} {{}}.", "a sculpture of {} {{}}.", "a photo of {} hard to see {{}}.",
        "a low resolution photo of {} {{}}.", "a rendering of {} {{}}.", "graffiti of {} {{}}.",
        "a cropped photo of {} {{}}.", "a tattoo of {} {{}}.", "{} embroidered {{}}.",
        "a bright photo of {} {{}}.", "a photo of {} clean {{}}.", "a photo of {} dirty {{}}.",
        "a dark photo of {} {{}}.", "a drawing of {} {{}}.", "{} plastic {{}}.", "a photo of {} cool {{}}.",
        "a close-up photo of {} {{}}.", "a black and white photo of {} {{}}.", "a painting of {} {{}}.",
        "a pixelated photo of {} {{}}.", "a jpeg corrupted photo of {} {{}}.", "a blurry photo of {} {{}}.",
        "a photo of {} {{}}.", "a good photo of {} {{}}.", "{} {{}} in a video game.", "a doodle of {} {{}}.",
        "{} origami {{}}.", "a sketch of {} {{}}.", "{} toy {{}}.", "a rendition of {} {{}}.",
        "a photo of {} large {{}}.", "a photo of {} nice {{}}.", "a photo of {} weird {{}}.", "{} cartoon {{}}.",
        "art of {} {{}}.", "{} plushie {{}}.", "a photo of {} small {{}}.", "itap of {} {{}}.",
    ]
    out = list(singles)
    for pat in paired:
        out.extend(a_the(pat))
    return out


IMAGENET_PROMPT_TEMPLATES = _imagenet_templates()


# --------------------------------------------------------------------------------------------
# ClipFusion
# --------------------------------------------------------------------------------------------


def _lazy_clip_features(fusion, rgb_imgs):
    """(fn, (C, npy, npx)) if the CLIP feature maps of ``rgb_imgs`` [B,H,W,3] may be computed later, in one backbone batch
    with the other frames queued behind ``integrate()``: only for this package's own ``Clip`` (a pure function of the image;
    an injected object may depend on when it is called), and only if the tiling fits (else the call raises now, as the
    reference does).  ``defer_backbone=False`` in the constructor switches it off."""
    clip = fusion.clip
    if not isinstance(clip, Clip) or not fusion.__dict__.get("defer_backbone", True) or not fusion.__dict__.get("defer_frames", True):
        return None
    ps, stride = int(fusion.clip_patch_size), int(fusion.clip_patch_stride)
    h, w = int(rgb_imgs.shape[1]), int(rgb_imgs.shape[2])
    if rgb_imgs.dim() != 4 or h < ps or w < ps or (h - ps) % stride != 0 or (w - ps) % stride != 0:
        return None
    fn = fusion.__dict__.get("_lazy_fn")
    if fn is None or fn[1:] != (ps, stride, id(clip)):
        f = lambda rgb_nhwc: clip.img_inference_tiled(rgb_nhwc.permute(0, 3, 1, 2), patch_size=ps, patch_stride=stride)
        fn = (f, ps, stride, id(clip))
        fusion.__dict__["_lazy_fn"] = fn  # one function object per configuration: the queue's key compares its id
    return fn[0], (int(clip.feature_dim), 1 + (h - ps) // stride, 1 + (w - ps) // stride)


class ClipFusion(_FusionVolumeMixin, torch.nn.Module):
    """Dense voxel volume with projective running-average fusion (reference clipfusion.py:575-763).

    Constructor, attributes and registered buffer names follow the reference; ``integrate`` runs
    the hand-written HIP path (include/saf.h: saf_fuse_frames).  ``clip`` may be passed as a ready
    ``Clip``-like object through ``clip_model`` (anything with ``feature_dim`` and
    ``img_inference_tiled``), which is how tests and the benchmark inject seeded feature maps.
    """

    _rgb_bilinear = False  # nearest rgb sampling (clipfusion.py:701-706)

    def __init__(self, origin, voxel_size, nvox, trunc, scale_patches_by_depth, clip_model, clip_pretraining,
                 clip_patch_size, clip_patch_stride, keep_xyz_world=True, feat_dtype=torch.float32, defer_frames=True,
                 index_offset=(0, 0, 0), x_planes=None, defer_backbone=True, device=None):
        super().__init__()
        self.__dict__["defer_frames"] = bool(defer_frames)
        self.__dict__["defer_backbone"] = bool(defer_backbone)
        if isinstance(clip_model, str):
            self.clip = Clip(clip_model, clip_pretraining)
            self.clip.requires_grad_(False)
            self.clip.eval()
        else:
            self.clip = clip_model
        self.clip_patch_size = clip_patch_size
        self.clip_patch_stride = clip_patch_stride
        self.scale_patches_by_depth = scale_patches_by_depth
        self._init_volume(origin, voxel_size, nvox, trunc, self.clip.feature_dim, 0, keep_xyz_world, feat_dtype, index_offset,
                          x_planes, device)

    def integrate(self, depth_imgs, rgb_imgs, poses, K):
        """Fuse a batch of frames (reference clipfusion.py:627-721).  Batch elements are folded in
        one after the other; for B > 1 the reference updates the TSDF jointly over the batch
        (:681-695), which is the same mean up to fp32 rounding."""
        if self.scale_patches_by_depth:
            clip_feat_img = self.clip.img_inference_tiled_depthscaled(
                rgb_imgs.permute(0, 3, 1, 2), depth_imgs, K, patch_stride=self.clip_patch_stride
            )
        else:
            lazy = _lazy_clip_features(self, rgb_imgs)
            if lazy is not None:
                return self._fuse(depth_imgs, rgb_imgs, poses, K, None, None, False, lazy_feat=lazy)
            clip_feat_img = self.clip.img_inference_tiled(
                rgb_imgs.permute(0, 3, 1, 2), patch_size=self.clip_patch_size, patch_stride=self.clip_patch_stride
            )
        self._fuse(depth_imgs, rgb_imgs, poses, K, clip_feat_img, None, False)


# --------------------------------------------------------------------------------------------
# backproject_pcd / bounds
# --------------------------------------------------------------------------------------------


def get_pix_vecs(imwidth, imheight, K):
    """Ray direction K^-1 [u,v,1]^T of every pixel, [B,H*W,3] (reference clipfusion.py:497-507).
    Kept for API compatibility; backproject_pcd below evaluates only the 7x7 lattice on the GPU."""
    v, u = torch.meshgrid(
        torch.arange(imheight, dtype=torch.float32, device=K.device),
        torch.arange(imwidth, dtype=torch.float32, device=K.device),
        indexing="ij",
    )
    uv1 = torch.stack((u, v, torch.ones_like(u)), dim=0).reshape(3, -1)
    return (K.inverse() @ uv1).transpose(1, 2)


def backproject_pcd(dataset, batch_size=1, num_workers=0, device="cpu", max_depth=torch.inf):
    """Sparse world-space point cloud of a scan: a 7x7 pixel lattice per frame un-projected with
    its depth (reference clipfusion.py:510-572).  The per-frame arithmetic runs in
    saf_backproject_lattice on the HIP device; ``device`` only says where the frames are staged
    in the reference and is accepted for compatibility.  Returns (xyz[M,3], rgb[M,3]) on the CPU."""
    if not torch.cuda.is_available():
        raise SafError("backproject_pcd needs the MI355X device; there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, num_workers=num_workers)
    uv_size = 7
    u = torch.round(torch.linspace(0, dataset.imwidth - 1, uv_size)).to(torch.int32)
    v = torch.round(torch.linspace(0, dataset.imheight - 1, uv_size)).to(torch.int32)
    u_dev, v_dev = u.to(dev), v.to(dev)
    vv, uu = torch.meshgrid(v.long(), u.long(), indexing="ij")
    vv, uu = vv.reshape(-1), uu.reshape(-1)
    npts = uv_size * uv_size
    xyz_dev, valid_dev, rgb_all = [], [], []
    # frames travel through two reusable pinned buffers (a copy from the loader's freshly allocated pageable batch costs
    # milliseconds on ROCm: the pages are pinned for the one transfer), results stay on the device until the end: one sync
    pinned, events, k = {}, [None, None], 0
    stream = torch.cuda.current_stream(dev)
    for rgb_imgs, depth_imgs, poses, K, _ in loader:
        bsz = len(rgb_imgs)
        if events[k] is not None:
            events[k].synchronize()
        staged = []
        for i, t in enumerate((depth_imgs.to(torch.float32), poses.to(torch.float32), K.to(torch.float32).inverse())):
            buf = pinned.get((k, i))
            if buf is None or buf.shape != t.shape:
                buf = pinned[(k, i)] = torch.empty(t.shape, dtype=torch.float32).pin_memory()
            buf.copy_(t)
            staged.append(buf.to(dev, non_blocking=True))
        ev = torch.cuda.Event()
        ev.record(stream)
        events[k], k = ev, 1 - k
        depth_d, poses_d, kinv_d = staged
        xyz = torch.empty((bsz, npts, 3), dtype=torch.float32, device=dev)
        valid = torch.empty((bsz, npts), dtype=torch.uint8, device=dev)
        for i in range(bsz):
            rc = lib().saf_backproject_lattice(
                depth_d[i].data_ptr(), dataset.imheight, dataset.imwidth, poses_d[i].data_ptr(), kinv_d[i].data_ptr(),
                u_dev.data_ptr(), uv_size, v_dev.data_ptr(), uv_size, float(max_depth), xyz[i].data_ptr(),
                valid[i].data_ptr(), stream.cuda_stream,
            )
            check(rc, "saf_backproject_lattice")
        for t in staged:
            t.record_stream(stream)
        xyz_dev.append(xyz)
        valid_dev.append(valid)
        rgb_all.append(rgb_imgs[:, vv, uu])
    if not xyz_dev:
        return torch.zeros((0, 3)), torch.zeros((0, 3))
    valid = torch.cat(valid_dev).bool().cpu()
    return torch.cat(xyz_dev).cpu()[valid], torch.cat(rgb_all)[valid]


def scene_bounds(xyz, voxel_size, trunc_m):
    """origin / nvox of the volume from the sparse cloud: 1st / 99th percentile -/+ trunc
    (reference clipfusion.py:1098-1106, clip_seem_fusion.py:278-287)."""
    pts = xyz.cpu().numpy() if isinstance(xyz, torch.Tensor) else np.asarray(xyz)
    minbound = torch.tensor(np.percentile(pts, 1, axis=0)).float() - trunc_m
    maxbound = torch.tensor(np.percentile(pts, 99, axis=0)).float() + trunc_m
    nvox = ((maxbound - minbound) / voxel_size).round().int()
    return minbound, nvox
