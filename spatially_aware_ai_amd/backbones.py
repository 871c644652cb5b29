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
