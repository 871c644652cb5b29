"""Host-side mirror of the reference's ``clip_seem_fusion.py`` hot-path API on MI355X.

  * ``ClipSeemFusion``        -- reference clip_seem_fusion.py:611-888: the ClipFusion volume plus a
                                 per-voxel panoptic class histogram, rgb sampled bilinearly.
  * ``argmax_with_check``     -- the manager's label decode (clip_seem_fusion.py:315-325).
  * ``discover_objects`` / ``extract_mesh_by_object`` -- the manager's object bookkeeping around the volume
                                 (handy_utils.py:295-480, :585-611).
  * ``TextQueryEngine``       -- the query half of ``InSituManager`` (``clip_text_query``,
                                 clip_seem_fusion.py:482-561) over fused vertex / voxel features.

The Flask app, datasets, flood fill and DGCNN parts of ``InSituManager`` are outside the fused hot
path; a maintainer keeps the reference's manager and swaps these classes in
(INTEGRATION.md).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _abi
from ._lib import SafError, check, current_stream_ptr, lib, require_cuda
from .clipfusion import _FusionVolumeMixin, _lazy_clip_features, _query_scan

N_PANOPTIC_SLOTS = 133 + 10  # clip_seem_fusion.py:655 -- 133 COCO panoptic classes, null = 133, spare slots


class ClipSeemFusion(_FusionVolumeMixin, torch.nn.Module):
    """CLIP + panoptic-label fusion volume (reference clip_seem_fusion.py:611-822).

    ``clip_model`` needs ``feature_dim`` and ``img_inference_tiled``; ``seg_model`` needs
    ``run_on_image(rgb[3,H,W]) -> class-id map [H,W]`` (kMaX-DeepLab in the reference,
    handy_utils.py:60-161).  Both run as PyTorch-ROCm modules; everything after them is HIP.
    Callers may set ``unique_objects``, ``voxel_obj_idx`` and ``objects_segmentation_color`` on the
    instance exactly as the reference's manager does (clip_seem_fusion.py:352-372).
    """

    _rgb_bilinear = True  # clip_seem_fusion.py:793-798

    def __init__(self, origin, voxel_size, nvox, trunc, scale_patches_by_depth, clip_patch_size, clip_patch_stride,
                 clip_model, seg_model, keep_xyz_world=True, feat_dtype=torch.float32, defer_frames=True,
                 index_offset=(0, 0, 0), x_planes=None, defer_backbone=True, device=None):
        super().__init__()
        self.__dict__["defer_frames"] = bool(defer_frames)
        self.__dict__["defer_backbone"] = bool(defer_backbone)
        self.clip = clip_model
        self.clip_patch_size = clip_patch_size
        self.clip_patch_stride = clip_patch_stride
        self.scale_patches_by_depth = scale_patches_by_depth
        self.segmentation_model = seg_model
        self.n_classes = N_PANOPTIC_SLOTS
        self._init_volume(origin, voxel_size, nvox, trunc, self.clip.feature_dim, self.n_classes, keep_xyz_world,
                          feat_dtype, index_offset, x_planes, device)
        self.debug_counter = 0

    def integrate(self, depth_imgs, rgb_imgs, poses, K):
        """Reference clip_seem_fusion.py:676-822.  (The reference indexes ``labels[i]`` on a
        size-1 dimension and therefore only works for B = 1; any B works here.)"""
        rgb_chw = rgb_imgs.permute(0, 3, 1, 2)
        if self.scale_patches_by_depth:
            clip_feat_img = self.clip.img_inference_tiled_depthscaled(
                rgb_chw, depth_imgs, K, patch_stride=self.clip_patch_stride
            )
        else:
            lazy = _lazy_clip_features(self, rgb_imgs)
            if lazy is not None:  # the ViT runs when the queue is flushed, on all queued frames at once
                label_maps = [self.segmentation_model.run_on_image(rgb_chw[i]).float().contiguous() for i in range(len(rgb_imgs))]
                return self._fuse(depth_imgs, rgb_imgs, poses, K, None, label_maps, True, lazy_feat=lazy)
            clip_feat_img = self.clip.img_inference_tiled(
                rgb_chw, patch_size=self.clip_patch_size, patch_stride=self.clip_patch_stride
            )
        label_maps = [self.segmentation_model.run_on_image(rgb_chw[i]).float() for i in range(len(rgb_imgs))]
        self._fuse(depth_imgs, rgb_imgs, poses, K, clip_feat_img, label_maps, True)

    def extract_mesh(self, marching_cubes=None):
        """Reference clip_seem_fusion.py:824-888: the 6-tuple (verts_world, faces, vertex_colors,
        vertex_clip_feats, vertex_obj_idx, vertex_segment_color); needs ``voxel_obj_idx`` and
        ``objects_segmentation_color`` set by the caller as the manager does (:352-372)."""
        verts, faces = self._marching_cubes_vertices(marching_cubes)
        colors, feats, obj, seg = self.sample_mesh_vertices(verts, self.voxel_obj_idx, self.objects_segmentation_color)
        return self._verts_world(verts), faces, colors, feats, obj, seg

    def label_index(self):
        """Per-voxel class id, -1 where nothing was fused: the manager's
        ``argmax_with_check_2d_efficient(labels_one_hot)`` (clip_seem_fusion.py:315-325)."""
        return argmax_with_check(self.labels_one_hot)


def argmax_with_check(labels_one_hot):
    """argmax over classes with all-zero rows -> -1 (clip_seem_fusion.py:315-320), int64 like
    torch.argmax."""
    require_cuda(labels_one_hot, "labels_one_hot")
    lab = labels_one_hot.contiguous()
    if lab.dtype != torch.int32:
        lab = lab.to(torch.int32)
    out = torch.empty(lab.shape[0], dtype=torch.int32, device=lab.device)
    with torch.cuda.device(lab.device):
        rc = lib().saf_label_argmax(lab.data_ptr(), lab.shape[0], lab.shape[1], out.data_ptr(), current_stream_ptr())
    check(rc, "saf_label_argmax")
    return out.long()


def label_components(labels_grid, null_class=133, min_voxels=3):
    """The object discovery of ``flood_fill_3d`` (handy_utils.py:295-480, no trained in-situ model) on the
    HIP device: 26-connected components of equal class over the label grid ``[nx,ny,nz]`` (the reshaped
    output of ``argmax_with_check``, clip_seem_fusion.py:322-338), the null class and empty voxels (-1)
    excluded, objects of fewer than ``min_voxels`` voxels rejected.

    Returns ``(voxel_obj_idx int32[nx,ny,nz], first_voxel int32[K], class_id int32[K], count int32[K])``:
    ``voxel_obj_idx`` is the reference's ``voxel_obj_ids`` (-1, or -2 - k for the k-th object in raster
    order of its first voxel, handy_utils.py:352-353, :447-450)."""
    require_cuda(labels_grid, "labels_grid")
    if labels_grid.dim() != 3:
        raise SafError(f"labels_grid must be [nx,ny,nz], got {tuple(labels_grid.shape)}")
    lab = labels_grid.to(torch.int32).contiguous()
    nx, ny, nz = (int(v) for v in lab.shape)
    n = lab.numel()
    dev = lab.device
    L = lib()
    ws = torch.empty(L.saf_label_components_workspace_bytes(n), dtype=torch.uint8, device=dev)
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    nobj = torch.zeros(1, dtype=torch.int32, device=dev)
    # an object has >= min_voxels voxels: at most n // min_voxels objects
    cap = max(1, n // max(1, int(min_voxels)))
    first = torch.empty(cap, dtype=torch.int32, device=dev)
    cls = torch.empty(cap, dtype=torch.int32, device=dev)
    cnt = torch.empty(cap, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.saf_label_components(lab.data_ptr(), nx, ny, nz, int(null_class), int(min_voxels), ids.data_ptr(),
                                    nobj.data_ptr(), cap, first.data_ptr(), cls.data_ptr(), cnt.data_ptr(),
                                    ws.data_ptr(), ws.numel(), current_stream_ptr())
    check(rc, "saf_label_components")
    k = int(nobj.item())
    return ids.view(nx, ny, nz), first[:k], cls[:k], cnt[:k]


def _obj_counts(object_counts, obj_id):
    """handy_utils.py:483-498: "<label>:<n>" ids count under their label; the running count makes the new id."""
    if ":" in obj_id:
        possible_label, possible_int = obj_id.split(":")[0], obj_id.split(":")[-1]
        if possible_int.isdigit():
            class_label = possible_label
        else:  # the reference dies here with an UnboundLocalError; keep the whole string as the label instead
            class_label = obj_id
    else:
        class_label = obj_id
    object_counts[class_label] = object_counts.get(class_label, 0) + 1
    return f"{class_label}:{object_counts[class_label]}", class_label


def discover_objects(labels_grid, class_names, class_colors=None, null_class=133, min_voxels=3, insitu_model=None,
                     voxel_clip_feats=None, voxel_rgb=None, scene_knowledge_prev=None, preprocess=None, arrays=False):
    """``scene_knowledge`` and ``voxel_obj_idx`` as ``flood_fill_3d`` builds them (handy_utils.py:295-480).

    The connected components come from the HIP kernel (``label_components``); objects are visited in raster order of
    their first voxel, which is the reference's discovery order.

    * First scan (``insitu_model`` None or not ``model_trained``): ids ``"<class label>:<running count>"``, indices
      ``-2, -3, ...`` (handy_utils.py:352-353, :430-452).
    * Repeat scan (handy_utils.py:396-452, :455-478): for every object its voxels' features (``voxel_clip_feats``
      [nx,ny,nz,D], ``voxel_rgb`` [nx,ny,nz,3], coordinates) go through ``preprocess([features], None, inference=True)``
      (the reference's ``InSituVoxelData.preprocess``; default: the feature dict itself in a list) to
      ``insitu_model.predict``; a prediction > 0 re-identifies the object: it takes the user's label
      ``insitu_model.labels[pred]``, the POSITIVE voxel index ``pred``, ``user_modified``, is listed in
      ``unchanged_objects``, and its id joins ``insitu_model.labels`` if new; other objects keep running negative
      indices.  Labels of the previous scan that found no object go to ``missing_objects`` (needs
      ``scene_knowledge_prev``).  The DGCNN classifier itself is the reference's (dgcnn/, not part of the fused path):
      any object with ``model_trained``, ``labels`` and ``predict`` works.

    ``voxels`` holds each object's voxel coordinates in raster order (the reference lists them in flood-fill visiting
    order) -- a list of tuples of Python ints as ``flood_fill_3d`` builds it, or, with ``arrays=True``, an ``io.ArrayList``
    around the [n, 3] coordinate array (reads like that list; no Python object per voxel: 0.4 s per scan of the reference's
    largest grid).  Returns ``(scene_knowledge, voxel_obj_idx int32 [nx,ny,nz])``."""
    from .io import ArrayList

    comp_idx, first, cls, cnt = label_components(labels_grid, null_class, min_voxels)
    unique_objects, object_counts = {}, {}
    unchanged_objects, new_objects, missing_objects = {}, {}, {}
    flat = comp_idx.reshape(-1)
    order = torch.argsort(-flat.long(), stable=True)  # -1 (none) first, then object 0, 1, ... each in raster order
    n_none = int((flat == -1).sum())
    order = order[n_none:]
    ny, nz = int(labels_grid.shape[1]), int(labels_grid.shape[2])
    coords_dev = torch.stack((order // (ny * nz), (order // nz) % ny, order % nz), dim=1)
    coords = coords_dev.cpu().numpy()
    offs = np.concatenate(([0], np.cumsum(cnt.cpu().numpy())))
    cls_h = cls.cpu().tolist()
    trained = bool(insitu_model is not None and insitu_model.model_trained)
    labels_frozen = list(insitu_model.labels[1:]) if insitu_model is not None else []  # handy_utils.py:364
    gt_labels = insitu_model.labels if insitu_model is not None else ["null"]
    if trained and (voxel_clip_feats is None or voxel_rgb is None):
        raise ValueError("a trained in-situ model needs voxel_clip_feats and voxel_rgb to describe the objects")
    if trained:
        vcf = torch.as_tensor(voxel_clip_feats)
        vcf = vcf.reshape(-1, vcf.shape[-1])
        vrgb = torch.as_tensor(voxel_rgb).reshape(-1, 3)
    new_index = torch.empty(len(cls_h), dtype=torch.int32)  # voxel index of object k in the final grid
    negative_object_index = -2
    for k, class_id in enumerate(cls_h):
        class_label = class_names[class_id]
        user_modified = False
        object_index = negative_object_index
        pred = 0
        if trained:
            sel = order[offs[k]:offs[k + 1]]
            feats = {"clip_feats": vcf[sel.to(vcf.device)], "rgb": vrgb[sel.to(vrgb.device)],
                     "voxels": coords[offs[k]:offs[k + 1]]}
            all_features = preprocess([feats], None, inference=True) if preprocess is not None else [feats]
            pred = int(insitu_model.predict(all_features))
            if pred > 0:  # found in the previous scan: the user's label and a stable, positive index
                class_label = insitu_model.labels[pred]
                user_modified = True
                object_index = pred
        obj_id, class_label = _obj_counts(object_counts, class_label)
        if user_modified and obj_id not in gt_labels:
            gt_labels.append(obj_id)  # handy_utils.py:268-272: training labels are defined here
        unique_objects[obj_id] = {
            "class_id": class_id,
            "class_label": class_label,
            # (tuples of Python ints, as flood_fill_3d lists them -- or the coordinate array standing in for that list)
            "voxels": ArrayList(coords[offs[k]:offs[k + 1]], tuples=True) if arrays else list(map(tuple, coords[offs[k]:offs[k + 1]].tolist())),
            "object_index": object_index,
            "gt_label": obj_id,
            "user_modified": user_modified,
            "merged": "merged" in class_label,
            "removed": False,
            "color": None if class_colors is None else class_colors[class_id],
        }
        if trained and pred > 0:
            unchanged_objects[obj_id] = unique_objects[obj_id]
        new_index[k] = object_index
        if object_index < 0:
            negative_object_index -= 1
    if scene_knowledge_prev:
        for gt_label in labels_frozen:
            if gt_label not in unique_objects:
                missing_objects[gt_label] = scene_knowledge_prev["unique_objects"][gt_label]
    if len(cls_h):
        # the kernel numbered the objects -2 - k; re-identified objects carry their positive label index instead
        lut = new_index.to(comp_idx.device)
        voxel_obj_idx = torch.where(comp_idx < -1, lut[(-2 - comp_idx).clamp(min=0).long()], comp_idx)
    else:
        voxel_obj_idx = comp_idx
    scene_knowledge = {"unique_objects": unique_objects, "object_counts": object_counts,
                       "unchanged_objects": unchanged_objects, "new_objects": new_objects, "missing_objects": missing_objects}
    return scene_knowledge, voxel_obj_idx


def extract_mesh_by_object(vertices, faces, colors, vertex_indices, obj_idx):
    """The sub-mesh of one object (handy_utils.py:585-611): the vertices whose sampled object index equals ``obj_idx``,
    the faces all of whose corners are such vertices, re-indexed.  Returns ``(object_vertices, object_faces,
    object_colors, mesh)``; ``mesh`` is an ``open3d.geometry.TriangleMesh`` where open3d is installed (the reference
    builds one and its caller ignores it, clip_seem_fusion.py:396-402), else None.  The re-indexing is a lookup table
    instead of the reference's Python loop over faces."""
    vertices, faces, colors = np.asarray(vertices), np.asarray(faces), np.asarray(colors)
    vi = np.asarray(vertex_indices)
    object_indices = np.where(vi.reshape(len(vi), -1)[:, 0] == obj_idx)[0]
    object_vertices, object_colors = vertices[object_indices], colors[object_indices]
    lut = np.full(len(vertices), -1, dtype=np.int64)
    lut[object_indices] = np.arange(len(object_indices))
    mapped = lut[faces] if len(faces) else np.zeros((0, 3), np.int64)
    object_faces = mapped[(mapped >= 0).all(axis=1)].astype(faces.dtype if len(faces) else np.int64)
    mesh = None
    try:
        import open3d as o3d

        mesh = o3d.geometry.TriangleMesh()
        mesh.vertices = o3d.utility.Vector3dVector(object_vertices)
        mesh.triangles = o3d.utility.Vector3iVector(object_faces)
        mesh.vertex_colors = o3d.utility.Vector3dVector(object_colors)
    except ImportError:
        pass
    return object_vertices, object_faces, object_colors, mesh


class TextQueryEngine:
    """The natural-language query of ``InSituManager.clip_text_query`` (clip_seem_fusion.py:482-561)
    over a set of fused feature rows (mesh-vertex features in the reference, :423; raw voxel rows work
    the same way).

    ``clip_model`` supplies ``encode_text_with_prompt_ensemble``; features stay resident on the HIP
    device and every query is one fused scan (row-normalise + nan_to_num + surgery epilogue).
    """

    def __init__(self, clip_model, vert_clip_feat, verts=None, faces=None, scene_knowledge=None, device=None):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.clip_model = clip_model
        self.vert_clip_feat = torch.as_tensor(vert_clip_feat).to(device=device, dtype=torch.float32).contiguous()
        self.verts = verts
        self.faces = faces
        self.scene_knowledge = scene_knowledge
        self.control_objects = None
        self.control_text_features = None

    def _ensure_controls(self, text):
        if self.control_objects is None:
            uo = (self.scene_knowledge or {}).get("unique_objects", {})
            self.control_objects = list(set(uo[k]["class_label"] for k in uo.keys()))
        if text not in self.control_objects or self.control_text_features is None:
            self.control_objects.append(text)
            self.control_text_features = self.clip_model.encode_text_with_prompt_ensemble(
                self.control_objects, "cpu", prompt_templates=["a photo of {}"]
            )

    def similarity(self, text):
        """[1, Nq, T] surgery similarity against the control set with ``text`` appended."""
        self._ensure_controls(text)
        return _query_scan(self.vert_clip_feat, self.control_text_features, _abi.SAF_Q_SURGERY, normalize=True)[None]

    def relevance(self, text):
        """Min-max normalised relevance in [0,1] per row (clip_seem_fusion.py:527-533), numpy f32."""
        sim = self.similarity(text)
        for n, name in enumerate(self.control_objects):
            if name != text:
                continue
            rel = sim[0, :, n].cpu().numpy()
            rel -= rel.mean()
            rel = np.clip(rel, 0, 1)
            return (rel - rel.min()) / (rel.max() - rel.min())
        return None

    def clip_text_query(self, text: str):
        """-> {"vertices","faces","colors"(RGBA)} or None, as the reference returns to /text_query."""
        rel = self.relevance(text)
        if rel is None:
            return None
        return {"vertices": self.verts, "faces": self.faces, "colors": relevance_to_rgba(rel).tolist()}

    def clip_text_query_json(self, text: str):
        """The same answer already serialised (UTF-8 JSON bytes, or None): what app_unity.py:66-71 hands to jsonify, without
        the detour through Python lists (io.mesh_to_json; vertices and faces parse to exactly clip_text_query's lists, the RGBA values -- float64 out of matplotlib -- are rounded to f32 first)."""
        from .io import mesh_to_json

        rel = self.relevance(text)
        if rel is None:
            return None
        return mesh_to_json(self.verts, self.faces, relevance_to_rgba(rel).astype(np.float32))


def relevance_to_rgba(relevance):
    """turbo colour map, alpha = 0.5 * relevance (clip_seem_fusion.py:544-548)."""
    import matplotlib

    rgb = matplotlib.colormaps["turbo"](relevance)[:, :3]
    alpha = relevance * 0.5
    return np.hstack([rgb, alpha[:, None]])
