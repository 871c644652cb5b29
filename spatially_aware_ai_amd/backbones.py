"""Stand-in backbones with the interfaces the fusion classes expect (PyTorch-ROCm; GEMMs on
hipBLASLt / MFMA).  They exist because neither ``open_clip`` / ``detectron2`` nor any pretrained
weights are available offline: the fused path treats backbone outputs as inputs, so for end-to-end
timing the *shape* of the work is what matters (SURVEY.md §7 "Backbones are unavailable offline").

  * ``RandomViTB32``  -- the architecture of open_clip's ViT-B/32 image tower (patch 32, width 768,
                         12 layers, 12 heads, QuickGELU, 512-d projection) with seeded random weights,
                         exposing ``visual.output_dim`` and ``encode_image`` like the object
                         ``open_clip.create_model`` returns (reference clipfusion.py:769-781, :833).
  * ``RandomPanoptic`` -- a constant-time stand-in for ``KmaxSegmentationModel.run_on_image``
                         (handy_utils.py:60-161): class-id map [H, W] from a tiny strided conv.
  * ``RandomKmaxConvNeXtL`` -- the SHAPE of the reference's kMaX-DeepLab panoptic model (handy_utils.py:29-58: detectron2
                         ``build_model`` of the kMaX ConvNeXt-L config) behind detectron2's calling convention, with seeded
                         random weights: a ConvNeXt-L encoder (depths 3-3-27-3, widths 192-384-768-1536, 7x7 depthwise
                         convolutions, at the 1281 x 960 input ``KmaxSegmentationModel.preprocess`` produces) and a light
                         stand-in for the kMaX decoder (lateral 1x1 convolutions, top-down sum, 128 mask queries, per-pixel
                         argmax).  Injected into ``segmentation.KmaxSegmentationModel`` it makes BASELINE config 3 -- kMaX +
                         CLIP jointly fused -- measurable end to end; the encoder carries the work (about 0.84 TFLOP per frame).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Block(nn.Module):
    def __init__(self, width: int, heads: int):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.qkv = nn.Linear(width, 3 * width)
        self.out = nn.Linear(width, width)
        self.ln_2 = nn.LayerNorm(width)
        self.fc1 = nn.Linear(width, 4 * width)
        self.fc2 = nn.Linear(4 * width, width)
        self.heads = heads

    def forward(self, x):  # [B, T, C]
        b, t, c = x.shape
        q, k, v = self.qkv(self.ln_1(x)).view(b, t, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, t, c)
        x = x + self.out(a)
        h = self.fc1(self.ln_2(x))
        return x + self.fc2(h * torch.sigmoid(1.702 * h))  # QuickGELU


class _VisualTower(nn.Module):
    def __init__(self, width=768, layers=12, heads=12, patch=32, image=224, output_dim=512):
        super().__init__()
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(3, width, patch, patch, bias=False)
        n_tok = (image // patch) ** 2 + 1
        self.class_embedding = nn.Parameter(torch.zeros(width))
        self.positional_embedding = nn.Parameter(torch.zeros(n_tok, width))
        self.ln_pre = nn.LayerNorm(width)
        self.blocks = nn.ModuleList(_Block(width, heads) for _ in range(layers))
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(torch.zeros(width, output_dim))

    def forward(self, x):  # [B, 3, 224, 224]
        x = self.conv1(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.class_embedding.expand(x.shape[0], 1, -1).to(x.dtype), x], dim=1)
        x = self.ln_pre(x + self.positional_embedding.to(x.dtype))
        for blk in self.blocks:
            x = blk(x)
        return self.ln_post(x[:, 0]) @ self.proj.to(x.dtype)


class RandomViTB32(nn.Module):
    """ViT-B/32-shaped image tower with seeded random weights (no text tower)."""

    def __init__(self, seed: int = 0, output_dim: int = 512):
        super().__init__()
        self.visual = _VisualTower(output_dim=output_dim)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g) * (p.shape[-1] ** -0.5))
                elif p.numel() and not p.is_floating_point():
                    continue
            for m in self.modules():
                if isinstance(m, nn.LayerNorm):
                    m.weight.fill_(1.0)
                    m.bias.zero_()

    def encode_image(self, x):
        return self.visual(x)

    def encode_text(self, tokens):
        raise NotImplementedError("the stand-in has no text tower; pass text features to the query functions")


class RandomPanoptic:
    """``run_on_image(rgb[3,H,W]) -> int64 class ids [H,W]`` in [0, 134), cheap and deterministic."""

    def __init__(self, n_classes: int = 134):
        self.n_classes = n_classes

    def run_on_image(self, rgb_chw):
        # a fixed hash of the quantised colours: stands in for the detectron2 / kMaX forward
        q = (rgb_chw * 255.0).to(torch.int64)
        return (q[0] * 7 + q[1] * 13 + q[2] * 29) % self.n_classes


class _LayerNorm2d(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.ln = nn.LayerNorm(c, eps=1e-6)

    def forward(self, x):  # [B, C, H, W]
        return self.ln(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


def dwconv7x7(x, weight, bias=None, _cache=None):
    """``F.conv2d(x, weight, bias, padding=3, groups=C)`` for a 7 x 7 depthwise convolution on a channels-last HIP tensor
    (``saf_dwconv7x7_nhwc``, csrc/saf_dwconv.hip: MIOpen runs this shape through naive_conv -- 46 % of the default bench's
    kernel time in round 3).  fp32 accumulation, output in x's dtype and memory format.  ``_cache``: a dict that keeps the
    re-laid [7, 7, C] f32 weight across calls (keyed by the weight's version)."""
    import ctypes as C

    from . import _abi
    from ._lib import check, current_stream_ptr, lib

    n, c, h, w = x.shape
    dt = {torch.float32: _abi.SAF_F32, torch.bfloat16: _abi.SAF_BF16, torch.float16: _abi.SAF_F16}[x.dtype]
    key = (weight.data_ptr(), weight._version, x.device) + ((None, None) if bias is None else (bias.data_ptr(), bias._version))
    if _cache is not None and _cache.get("key") == key:
        w_kkc, b32 = _cache["w"], _cache["b"]
    else:
        w_kkc = weight.detach()[:, 0].permute(1, 2, 0).contiguous().to(device=x.device, dtype=torch.float32)
        b32 = None if bias is None else bias.detach().to(device=x.device, dtype=torch.float32).contiguous()
        if _cache is not None:
            _cache.update(key=key, w=w_kkc, b=b32)
    y = torch.empty_like(x, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        check(lib().saf_dwconv7x7_nhwc(x.data_ptr(), w_kkc.data_ptr(), None if b32 is None else b32.data_ptr(), y.data_ptr(),
                                       n, h, w, c, dt, current_stream_ptr()), "saf_dwconv7x7_nhwc")
    return y


def _dwconv_ok(x, conv):
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and conv.kernel_size == (7, 7) and conv.padding == (3, 3)
            and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == x.shape[1] == conv.out_channels
            # inference only (no autograd through the HIP op)
            and not (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)))


class _ConvNeXtBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dw = nn.Conv2d(c, c, 7, padding=3, groups=c)
        self.ln = nn.LayerNorm(c, eps=1e-6)
        self.fc1 = nn.Linear(c, 4 * c)
        self.fc2 = nn.Linear(4 * c, c)
        self.gamma = nn.Parameter(torch.full((c,), 1e-6))
        self.__dict__["_dw_cache"] = {}

    def _depthwise(self, x):
        # under autocast the library convolution would compute in the autocast dtype: so does this one
        if torch.is_autocast_enabled("cuda") and x.is_cuda and x.dtype == torch.float32:
            x = x.to(torch.get_autocast_dtype("cuda"))
        if _dwconv_ok(x, self.dw):
            return dwconv7x7(x, self.dw.weight, self.dw.bias, self.__dict__["_dw_cache"])
        return self.dw(x)

    def forward(self, x):
        y = self._depthwise(x).permute(0, 2, 3, 1)
        y = self.fc2(F.gelu(self.fc1(self.ln(y)))) * self.gamma
        return x + y.permute(0, 3, 1, 2)


class RandomKmaxConvNeXtL(nn.Module):
    """``model([{"image": int32 BGR [3,h,w], "height": H, "width": W}]) -> [{"panoptic_seg": (ids [H,W], segments_info)}]``:
    detectron2's interface for the kMaX-DeepLab ConvNeXt-L panoptic model, random weights (see the module docstring)."""

    DEPTHS, DIMS, QUERIES, EMBED = (3, 3, 27, 3), (192, 384, 768, 1536), 128, 128

    def __init__(self, seed: int = 0, n_classes: int = 133):
        super().__init__()
        d = self.DIMS
        self.stem = nn.Sequential(nn.Conv2d(3, d[0], 4, 4), _LayerNorm2d(d[0]))
        self.stages = nn.ModuleList()
        self.down = nn.ModuleList()
        for i, (n, c) in enumerate(zip(self.DEPTHS, d)):
            if i:
                self.down.append(nn.Sequential(_LayerNorm2d(d[i - 1]), nn.Conv2d(d[i - 1], c, 2, 2)))
            self.stages.append(nn.Sequential(*[_ConvNeXtBlock(c) for _ in range(n)]))
        self.lateral = nn.ModuleList(nn.Conv2d(c, 256, 1) for c in d)
        self.pixel = nn.Conv2d(256, self.EMBED, 3, padding=1)
        self.centers = nn.Parameter(torch.zeros(self.QUERIES, self.EMBED))
        self.register_buffer("pixel_mean", torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1))
        self.register_buffer("pixel_std", torch.tensor([57.375, 57.12, 58.395]).view(3, 1, 1))
        self.n_classes = n_classes
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() > 1:
                    fan_in = p[0].numel()
                    p.copy_(torch.randn(p.shape, generator=g) * fan_in ** -0.5)
            for m in self.modules():
                if isinstance(m, nn.LayerNorm):
                    m.weight.fill_(1.0)
                    m.bias.zero_()
                elif isinstance(m, _ConvNeXtBlock):
                    m.gamma.fill_(0.1)
        self.query_class = [(7 * q + 3) % n_classes for q in range(self.QUERIES)]

    def forward(self, batched_inputs):
        out = []
        for inp in batched_inputs:
            img = inp["image"].to(self.pixel_mean.device).float()
            x = ((img - self.pixel_mean) / self.pixel_std)[None].contiguous(memory_format=torch.channels_last)
            feats = []
            x = self.stem(x)
            for i, stage in enumerate(self.stages):
                if i:
                    x = self.down[i - 1](x)
                x = stage(x)
                feats.append(x)
            y = self.lateral[3](feats[3])
            for i in (2, 1, 0):  # top-down: the pixel decoder's place
                y = self.lateral[i](feats[i]) + F.interpolate(y, size=feats[i].shape[-2:], mode="nearest")
            emb = self.pixel(y)  # [1, E, h/4, w/4]
            logits = torch.einsum("qe,behw->bqhw", self.centers.to(emb.dtype), emb)
            ids = logits.argmax(dim=1, keepdim=True).float()
            ids = F.interpolate(ids, size=(int(inp["height"]), int(inp["width"])), mode="nearest")[0, 0].long() + 1
            present = torch.unique(ids).tolist()
            info = [{"id": int(i), "category_id": self.query_class[int(i) - 1], "isthing": self.query_class[int(i) - 1] < 80}
                    for i in present]
            out.append({"panoptic_seg": (ids, info)})
        return out
