"""MI355X-native multimodal 3D fusion: the hot path of cy-xu/spatially_aware_AI
(projective voxel fusion of CLIP feature maps + panoptic labels, and the CLIP-text query
scan) as hand-written HIP for gfx950 behind the reference's Python entry points.

The public names mirror the reference modules (clipfusion.py / clip_seem_fusion.py):
``ClipFusion``, ``ClipSeemFusion``, ``Clip``, ``backproject_pcd``, ``get_pix_vecs``.
They are imported lazily so that CPU-only tooling (synthetic inputs, the golden
generator) does not need the HIP library.
"""

__all__ = [
    "ClipFusion",
    "ClipSeemFusion",
    "Clip",
    "backproject_pcd",
    "get_pix_vecs",
    "scene_bounds",
    "label_components",
    "discover_objects",
]


def __getattr__(name):
    if name in ("ClipFusion", "Clip", "backproject_pcd", "get_pix_vecs", "scene_bounds"):
        from . import clipfusion as _m

        return getattr(_m, name)
    if name in ("ClipSeemFusion", "label_components", "discover_objects"):
        from . import clip_seem_fusion as _m

        return getattr(_m, name)
    raise AttributeError(name)
