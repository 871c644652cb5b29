"""Host-side mirror of the reference's kMaX-DeepLab wrapper (``KmaxSegmentationModel``,
handy_utils.py:29-161).  The backbone stays what it is in the reference -- a detectron2 panoptic model with the kMaX
config and weights (``kmax`` package and checkpoint are not part of the reference snapshot) -- and is INJECTED; what is
restated here is everything the reference does around it:

* pre-processing (handy_utils.py:75-100): long edge to 1281 px (bilinear, align_corners=False), RGB -> BGR, x 255,
  truncation to int32, the ``{"image", "height", "width"}`` input dict;
* post-processing (handy_utils.py:104-135): the panoptic id map painted with category ids -- id 0 becomes the null
  class 133, every segment listed in ``segments_info`` takes its ``category_id`` ("stuff" first, then "things", as
  detectron2 0.6's ``_PanopticPrediction`` iterates them; the masks are disjoint, so this is one table lookup),
  ids the model did not describe stay as they are.

``run_on_image(rgb[3,H,W]) -> class-id map [H,W]`` is what ``ClipSeemFusion.integrate`` calls per frame
(clip_seem_fusion.py:755-760); the map stays on the device of the model's output.
"""
from __future__ import annotations

import torch

NULL_CLASS = 133  # handy_utils.py:112
LONG_EDGE = 1281  # handy_utils.py:80-84


class KmaxSegmentationModel:
    """``model`` is the detectron2 module the reference builds with ``build_model(cfg)`` + ``DetectionCheckpointer``
    (handy_utils.py:44-51): called with ``[{"image": int32 BGR [3,h,w], "height": H, "width": W}]`` it returns
    ``[{"panoptic_seg": (ids [H,W], segments_info)}]``.  ``KmaxSegmentationModel.from_config`` builds it exactly as
    the reference does when detectron2 and kmax are importable."""

    def __init__(self, model, device="cpu"):
        self.model = model
        self.device = torch.device(device)
        self.cpu_device = torch.device("cpu")

    @classmethod
    def from_config(cls, config_file, weight_path, device="cpu"):
        try:
            from detectron2.checkpoint import DetectionCheckpointer
            from detectron2.config import get_cfg
            from detectron2.modeling import build_model
            from detectron2.projects.deeplab import add_deeplab_config
            from kmax.kmax_deeplab import add_kmax_deeplab_config
        except ImportError as e:  # same failure mode as the reference's module import
            raise ImportError("detectron2 and the kmax package are required to build the kMaX-DeepLab model "
                              "(or pass a ready model to KmaxSegmentationModel)") from e
        cfg = get_cfg()
        add_deeplab_config(cfg)
        add_kmax_deeplab_config(cfg)
        cfg.merge_from_file(config_file)
        cfg.freeze()
        model = build_model(cfg)
        model.eval()
        DetectionCheckpointer(model).load(weight_path)
        return cls(model, device)

    @staticmethod
    def preprocess(image):
        """[3,H,W] RGB in 0..1 -> the model's input dict (handy_utils.py:75-100)."""
        _, height, width = image.shape
        aspect_ratio = width / height
        if aspect_ratio > 1:
            new_width = LONG_EDGE
            new_height = int(new_width / aspect_ratio)
        else:
            new_height = LONG_EDGE
            new_width = int(new_height * aspect_ratio)
        x = torch.nn.functional.interpolate(image.unsqueeze(0), size=(new_height, new_width), mode="bilinear",
                                            align_corners=False)[0]
        x = x[[2, 1, 0], :, :]
        x = (x * 255).to(torch.int32)
        return {"image": x, "height": height, "width": width}

    @staticmethod
    def paint_categories(panoptic_seg, segments_info):
        """Panoptic id map -> per-pixel category id (handy_utils.py:104-135)."""
        ids = panoptic_seg.long()
        top = int(ids.max()) if ids.numel() else 0
        lut = torch.arange(max(top, 0) + 1, dtype=panoptic_seg.dtype, device=panoptic_seg.device)  # unknown ids stay
        if lut.numel():
            lut[0] = NULL_CLASS
        for s in segments_info:  # disjoint masks: the order in which the reference paints them cannot matter
            if 0 <= int(s["id"]) <= top:
                lut[int(s["id"])] = int(s["category_id"])
        neg = ids < 0
        out = lut[ids.clamp(min=0)]
        return torch.where(neg, panoptic_seg, out) if bool(neg.any()) else out

    def run_on_image(self, image):
        with torch.no_grad():
            predictions = self.model([self.preprocess(image)])[0]
        panoptic_seg, segments_info = predictions["panoptic_seg"]
        return self.paint_categories(panoptic_seg, segments_info)
