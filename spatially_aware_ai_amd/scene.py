"""The scene-level flow of the reference's manager on the MI355X path, stage by stage and timed.

``reconstruct_scene`` is ``InSituManager.run_clipfusion`` (reference clip_seem_fusion.py:247-437) in the reference's own
order -- bounds from a sparse back-projection, one ``integrate`` call per frame, label decode, object discovery, the
attributes the manager sets on the volume from outside, ``extract_mesh``'s 6-tuple, per-object meshes, artefacts on disk --
and ``SceneResult.text_query`` is ``InSituManager.clip_text_query`` (:482-561) over what the reconstruction left.  Every
stage is one of the package's existing entry points (nothing is computed here that the hot path does not already own);
what this module adds is the chain and a wall-clock split per stage, the counterpart of the reference's one published
performance statement ("within a few minutes after a user scans the environment", README.md:4).

The Flask app, the datasets' decoders and the DGCNN in-situ learner stay the reference's.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from .clip_seem_fusion import ClipSeemFusion, TextQueryEngine, discover_objects, extract_mesh_by_object
from .clipfusion import backproject_pcd, scene_bounds
from .io import ArrayList, dumps_scene_knowledge, save_npy, save_ply, save_scene_arrays


@dataclass
class SceneResult:
    """What ``run_clipfusion`` leaves on the manager (clip_seem_fusion.py:322-424), plus the stage clock."""

    fusion: ClipSeemFusion
    origin: torch.Tensor
    nvox: torch.Tensor
    xyz: torch.Tensor  # the sparse preview cloud (clip_seem_fusion.py:269-275)
    onehot_to_index: torch.Tensor  # [nx,ny,nz] int64 on the device
    scene_knowledge: dict
    voxel_obj_idx: torch.Tensor
    verts: np.ndarray
    faces: np.ndarray
    vertex_colors: torch.Tensor
    vert_clip_feat: torch.Tensor
    vertex_obj_idx: torch.Tensor
    segmentation_color: torch.Tensor
    paths: dict = field(default_factory=dict)
    seconds: dict = field(default_factory=dict)
    engine: TextQueryEngine | None = None

    def text_query(self, clip_model, text):
        """``clip_text_query`` (clip_seem_fusion.py:482-561): RGBA heat map over the scene mesh, or None."""
        t0 = time.perf_counter()
        if self.engine is None:
            self.engine = TextQueryEngine(clip_model, self.vert_clip_feat, verts=self.verts.tolist(), faces=self.faces.tolist(),
                                          scene_knowledge=self.scene_knowledge)
        out = self.engine.clip_text_query(text)
        torch.cuda.synchronize()
        self.seconds["text_query"] = self.seconds.get("text_query", 0.0) + time.perf_counter() - t0
        return out


class FrameStager:
    """Host -> device staging of the loader's frames through a few reusable pinned buffers.

    The reference's loop calls ``.to(device)`` on the four freshly collated tensors of every frame
    (clip_seem_fusion.py:305-311).  On ROCm a copy from pageable memory pins the source pages for the transfer, and the
    DataLoader frees the batch right after: measured 6.6 ms per 640 x 480 frame for the four copies (tools/probe_scene.py),
    against 0.07 ms for the fused path to consume the frame.  Here the batch is copied into one of ``slots`` pinned buffers
    (a host memcpy) and sent from there asynchronously on the current stream; a slot is reused once its last transfer has
    completed (an event per slot).  What ``integrate`` receives is the same tensors on the device."""

    def __init__(self, device, slots=4):
        self.device = torch.device(device)
        self.slots = int(slots)
        self._bufs = {}
        self._events = [None] * self.slots
        self._k = 0

    def __call__(self, *tensors):
        k = self._k
        self._k = (k + 1) % self.slots
        if self._events[k] is not None:
            self._events[k].synchronize()
        out = []
        for i, t in enumerate(tensors):
            key = (k, i, tuple(t.shape), t.dtype)
            buf = self._bufs.get(key)
            if buf is None:
                self._bufs = {kk: v for kk, v in self._bufs.items() if kk[:2] != (k, i)}  # another shape took this place
                buf = self._bufs[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
            buf.copy_(t)
            out.append(buf.to(self.device, non_blocking=True))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._events[k] = ev
        return out


class _Clock:
    def __init__(self):
        self.seconds = {}
        self._t = None

    def start(self):
        torch.cuda.synchronize()
        self._t = time.perf_counter()

    def lap(self, name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.seconds[name] = self.seconds.get(name, 0.0) + now - self._t
        self._t = now


def reconstruct_scene(dataset, config, clip_model, seg_model, class_names, class_colors=None, device="cuda", out_dir=None,
                      max_depth=4, scale_patches_by_depth=False, feat_dtype=torch.float32, num_workers=0,
                      object_meshes=True, marching_cubes=None, python_lists=False):
    """``InSituManager.run_clipfusion`` (clip_seem_fusion.py:247-437).

    ``dataset`` yields the reference loaders' 5-tuple ``(rgb[H,W,3], depth[H,W], pose[4,4], K[3,3], idx)`` and has
    ``imwidth`` / ``imheight``; ``config`` needs ``voxel_size``, ``trunc_vox``, ``clip_patch_size``,
    ``clip_patch_stride`` (the reference's config dict, clip_seem_fusion.py:63-94).  ``class_names`` /
    ``class_colors`` are kMaX's ``COCO_PANOPTIC_CLASSES`` / ``COCO_PANOPTIC_COLORS`` in the reference
    (handy_utils.py:23-26).  Returns a ``SceneResult``; ``result.seconds`` holds the wall clock of every stage.

    ``python_lists``: keep the bulky members of ``scene_knowledge`` -- every object's ``voxels`` and ``mesh`` -- as the nested
    Python lists the reference builds (handy_utils.py:430-452, clip_seem_fusion.py:393-417); by default they are
    ``io.ArrayList``s around the arrays (same reads, same JSON: 0.8 s of a 2 s scan were spent building those lists and
    encoding them).  With ``out_dir`` the volume's ``.npy`` files (3.4 GB at the reference's largest grid) are written by a
    second thread from the moment the fusion is complete, beside the object / mesh stages."""
    clk = _Clock()
    clk.start()
    # ---- scene bounds (clip_seem_fusion.py:266-287)
    xyz, _ = backproject_pcd(dataset, batch_size=1, num_workers=num_workers, device="cpu", max_depth=max_depth)
    trunc_m = config["trunc_vox"] * config["voxel_size"]
    origin, nvox = scene_bounds(xyz, config["voxel_size"], trunc_m)
    clk.lap("bounds")
    # ---- the volume and the fusion loop, one frame per call (clip_seem_fusion.py:291-313)
    # (the reference builds the module on the host and moves it, clip_seem_fusion.py:291-302: gigabytes of zeros through pageable
    #  memory -- 0.7 s at the reference's largest grid; `device=` lets the buffers be born on the device)
    fusion = ClipSeemFusion(origin, config["voxel_size"], nvox, trunc_m, scale_patches_by_depth, config["clip_patch_size"],
                            config["clip_patch_stride"], clip_model, seg_model, feat_dtype=feat_dtype, device=device).to(device)
    loader = torch.utils.data.DataLoader(dataset, batch_size=1, num_workers=num_workers)
    stage = FrameStager(device)
    host = {"loader": 0.0, "stage": 0.0, "integrate": 0.0}  # where the host's time in the loop goes (no device sync inside)
    t_prev = time.perf_counter()
    for rgb_imgs, depth_imgs, poses, K, _ in loader:
        t0 = time.perf_counter()
        depth_d, rgb_d, poses_d, k_d = stage(depth_imgs.float(), rgb_imgs.float(), poses.float(), K.float())
        t1 = time.perf_counter()
        fusion.integrate(depth_d, rgb_d, poses_d, k_d)
        t2 = time.perf_counter()
        host["loader"] += t0 - t_prev
        host["stage"] += t1 - t0
        host["integrate"] += t2 - t1
        t_prev = t2
    fusion.flush()
    clk.lap("fuse")
    clk.seconds["fuse_host_split"] = {k: round(v, 4) for k, v in host.items()}
    # ---- the volume's artefacts (save_files_and_broadcast, :566-581) start now, on a thread of their own: nothing below writes
    #      clip_feat or rgb (the lap above synchronised the device; ctypes releases the interpreter lock for the whole write)
    writer = None
    if out_dir is not None:
        import threading

        os.makedirs(out_dir, exist_ok=True)
        nx_, ny_, nz_ = (int(v) for v in fusion.nvox)
        vol_rgb, vol_feat = fusion.rgb.view(nx_, ny_, nz_, 3), fusion.clip_feat.view(nx_, ny_, nz_, -1)
        wpaths, werr = {}, []

        def _write_volume():
            try:
                t_w = time.perf_counter()
                with torch.cuda.device(vol_feat.device), torch.cuda.stream(torch.cuda.Stream(vol_feat.device)):
                    wpaths["voxel_rgb"] = save_npy(os.path.join(out_dir, "voxel_rgb.npy"), vol_rgb)
                    wpaths["voxel_clip_feats"] = save_npy(os.path.join(out_dir, "voxel_clip_feats.npy"), vol_feat)
                wpaths["_seconds"] = time.perf_counter() - t_w
            except BaseException as e:  # noqa: BLE001 -- re-raised by the main thread at the join
                werr.append(e)

        writer = threading.Thread(target=_write_volume, name="saf-scene-writer")
        writer.start()
    # ---- labels: argmax with the empty check (:315-333), on the device
    onehot_to_index = fusion.label_index().view(*[int(v) for v in fusion.nvox])
    clk.lap("label_argmax")
    # ---- objects (flood_fill_3d, :341-348) and the attributes the manager sets from outside (:351-372)
    scene_knowledge, voxel_obj_idx = discover_objects(onehot_to_index, class_names, class_colors, arrays=not python_lists)
    scene_knowledge["scan_version"] = 0
    fusion.unique_objects = scene_knowledge["unique_objects"]
    fusion.voxel_obj_idx = voxel_obj_idx
    seg_color = torch.clone(fusion.rgb).view(*[int(v) for v in fusion.nvox], -1)
    for obj_info in scene_knowledge["unique_objects"].values():
        if obj_info["color"] is None:
            continue
        vox = torch.as_tensor(np.asarray(obj_info["voxels"], dtype=np.int64), device=seg_color.device)
        color = torch.tensor(obj_info["color"]).float() / 255.0
        seg_color[vox[:, 0], vox[:, 1], vox[:, 2]] = color.to(seg_color.device)
    fusion.objects_segmentation_color = seg_color.view(-1, 3)
    clk.lap("objects")
    # ---- the scene mesh with per-vertex colour, feature, object index, segment colour (:377-385)
    verts, faces, vertex_colors, vertex_clip_feats, vertex_obj_idx, segmentation_color = fusion.extract_mesh(marching_cubes)
    clk.lap("extract_mesh")
    # ---- per-object meshes into the scene knowledge (:393-417)
    if object_meshes:
        vc_h, vo_h = vertex_colors.cpu().numpy(), vertex_obj_idx.cpu().numpy()
        for obj_key, obj_value in scene_knowledge["unique_objects"].items():
            ov, of_, oc, _ = extract_mesh_by_object(verts, faces, vc_h, vo_h, obj_value["object_index"])
            if python_lists:
                mesh = {"vertices": ov.tolist(), "faces": of_.tolist(), "colors": oc.tolist()}
            else:
                mesh = {"vertices": ArrayList(ov), "faces": ArrayList(of_), "colors": ArrayList(oc)}
            scene_knowledge["unique_objects"][obj_key]["mesh"] = None if len(of_) < 10 else mesh
        clk.lap("object_meshes")
    res = SceneResult(fusion, origin, nvox, xyz, onehot_to_index, scene_knowledge, voxel_obj_idx, verts, faces, vertex_colors,
                      vertex_clip_feats, vertex_obj_idx, segmentation_color)
    # ---- artefacts (save_files_and_broadcast, :563-607)
    if out_dir is not None:
        res.paths = {"vertex_clip_feats": save_npy(os.path.join(out_dir, "vertex_clip_feats.npy"), torch.as_tensor(vertex_clip_feats)),
                     "vertex_obj_idx": save_npy(os.path.join(out_dir, "vertex_obj_idx.npy"), torch.as_tensor(vertex_obj_idx))}
        res.paths["mesh_rgb"] = save_ply(os.path.join(out_dir, "mesh_rgb.ply"), verts, faces, vertex_colors)
        res.paths["mesh_segmentation"] = save_ply(os.path.join(out_dir, "mesh_segmentation.ply"), verts, faces, segmentation_color)
        res.paths["scene_knowledge"] = os.path.join(out_dir, "scene_knowledge.json")
        with open(res.paths["scene_knowledge"], "w") as f:
            # the same text as json.dump(scene_knowledge, f, default=str) (clip_seem_fusion.py:603-604) -- but json.dump
            # walks the object with the pure-Python encoder, a chunk at a time: 2.1 s for this scene's voxel lists and
            # meshes; dumps() hands the whole object to the C encoder: 0.5 s
            # and the bulky lists (voxels, per-object meshes) are arrays rendered natively: 0.05 s
            f.write(dumps_scene_knowledge(scene_knowledge))
        clk.lap("save")
        writer.join()
        if werr:
            raise werr[0]
        clk.seconds["volume_write_beside"] = round(wpaths.pop("_seconds"), 4)
        res.paths.update(wpaths)
        clk.lap("save_volume_wait")
    res.seconds = clk.seconds
    return res
