"""Seeded synthetic RGB-D frames for parity tests and the benchmark.

Everything here is generated on the CPU with an explicit ``torch.Generator`` so the same
seed gives the same frames in the golden generator (this container), in the tests and in
``bench.py`` (GPU box).  Nothing here is part of the fused hot path.

Scene: a cube of side ``side`` metres centred on the world origin, voxelised ``nvox`` per
axis; cameras sit on a sphere of radius ``radius`` and look at the origin with
right-down-forward camera axes (the convention of the reference's loaders,
clipfusion.py:308-312), poses are camera->world 4x4.  Intrinsics ``fx=fy=0.9*W``,
``cx=W/2``, ``cy=H/2``.  Depth is either (A) iid U(1.5,3.5) m per pixel -- the incoherent
worst case -- or (B) an analytic sphere-in-a-box scene (coherent shell).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class GridSpec:
    origin: torch.Tensor  # f32[3], CPU
    voxel_size: float
    nvox: torch.Tensor  # i32[3]
    trunc: float

    @property
    def n_voxels(self) -> int:
        return int(torch.prod(self.nvox.long()).item())


def make_grid(nvox, side: float = 2.56, trunc_vox: float = 3.0) -> GridSpec:
    """Cube of ``side`` metres centred at the origin; ``nvox`` may be an int or a 3-tuple
    (non-cubic grids keep the voxel size of the first axis)."""
    if isinstance(nvox, int):
        nvox = (nvox, nvox, nvox)
    voxel_size = side / float(nvox[0])
    origin = torch.tensor([-0.5 * voxel_size * n for n in nvox], dtype=torch.float32)
    return GridSpec(
        origin=origin,
        voxel_size=voxel_size,
        nvox=torch.tensor(nvox, dtype=torch.int32),
        trunc=trunc_vox * voxel_size,
    )


def look_at_pose(centre: torch.Tensor) -> torch.Tensor:
    """cam->world 4x4 (f32) for a camera at ``centre`` looking at the world origin."""
    c = centre.double()
    fwd = -c / c.norm()
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    if abs(float(fwd @ up)) > 0.999:
        up = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    pose = torch.eye(4, dtype=torch.float64)
    pose[:3, 0] = right
    pose[:3, 1] = down
    pose[:3, 2] = fwd
    pose[:3, 3] = c
    return pose.float()


def intrinsics(width: int, height: int) -> torch.Tensor:
    k = torch.eye(3, dtype=torch.float32)
    k[0, 0] = 0.9 * width
    k[1, 1] = 0.9 * width
    k[0, 2] = width / 2
    k[1, 2] = height / 2
    return k


def _analytic_depth(pose: torch.Tensor, k: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """Depth (camera z) of a sphere r=0.9 m inside an axis-aligned box of half-size 1.2 m.  Works on
    whatever device ``pose`` lives on (the benchmark generates its frames on the GPU)."""
    dev = pose.device
    u = torch.arange(width, dtype=torch.float64, device=dev)
    v = torch.arange(height, dtype=torch.float64, device=dev)
    vv, uu = torch.meshgrid(v, u, indexing="ij")
    kd = k.double()
    rays_cam = torch.stack(
        ((uu - kd[0, 2]) / kd[0, 0], (vv - kd[1, 2]) / kd[1, 1], torch.ones_like(uu)), dim=-1
    )
    rot = pose[:3, :3].double()
    o = pose[:3, 3].double()
    d = rays_cam @ rot.T  # world-space ray per unit camera z
    # sphere |o + s d| = 0.9
    a = (d * d).sum(-1)
    b = 2 * (d @ o)
    c = (o @ o) - 0.9**2
    disc = b * b - 4 * a * c
    inf = torch.full_like(a, math.inf)
    s_sphere = torch.where(disc > 0, (-b - disc.clamp_min(0).sqrt()) / (2 * a), inf)
    s_sphere = torch.where(s_sphere > 0, s_sphere, inf)
    # box exit (camera is outside the box: take the far faces, i.e. the inside of the room)
    half = 1.2
    inv = 1.0 / d
    t1 = (-half - o) * inv
    t2 = (half - o) * inv
    s_box = torch.maximum(t1, t2).min(dim=-1).values
    s = torch.minimum(s_sphere, s_box)
    return s.float()


def make_frame(
    gen: torch.Generator,
    width: int,
    height: int,
    feat_dim: int,
    npy: int,
    npx: int,
    depth_kind: str = "A",
    radius: float = 2.5,
    n_label_classes: int = 134,
    missing_depth_frac: float = 0.0,
):
    """One frame: dict of CPU f32 tensors shaped like a B=1 batch of the reference's loaders
    (clipfusion.py:190) plus the per-frame feature map and label map that stand in for the
    CLIP / kMaX backbones."""
    c = torch.randn(3, generator=gen, dtype=torch.float32)
    c = c / c.norm() * radius
    pose = look_at_pose(c)
    k = intrinsics(width, height)
    if depth_kind == "A":
        depth = torch.rand(height, width, generator=gen, dtype=torch.float32) * 2.0 + 1.5
    elif depth_kind == "B":
        depth = _analytic_depth(pose, k, width, height)
    else:
        raise ValueError(depth_kind)
    if missing_depth_frac > 0:
        hole = torch.rand(height, width, generator=gen) < missing_depth_frac
        depth = depth.masked_fill(hole, 0.0)
    rgb = torch.rand(height, width, 3, generator=gen, dtype=torch.float32)
    feat = torch.randn(feat_dim, npy, npx, generator=gen, dtype=torch.float32)
    labels = torch.randint(0, n_label_classes, (height, width), generator=gen, dtype=torch.int64)
    return {
        "depth": depth[None],
        "rgb": rgb[None],
        "pose": pose[None],
        "K": k[None],
        "feat": feat[None],
        "labels": labels,
    }


def make_frames(seed: int, n_frames: int, **kw):
    gen = torch.Generator().manual_seed(seed)
    return [make_frame(gen, **kw) for _ in range(n_frames)]


def feature_map_shape(width: int, height: int):
    """Patch size H/3, stride H/6 (clipfusion.py:1199-1201 uses 160/80 at 640x480)."""
    p, s = height // 3, height // 6
    return (height - p) // s + 1, (width - p) // s + 1
