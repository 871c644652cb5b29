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
import zlib
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


# ---- the camera family real scans have (clipfusion.py:308-312: an ARKit pose with its y and z columns negated -- arbitrary roll
# and pitch; clipfusion.py:647-659: K @ xyz_cam with the dataset's own fx != fy and principal point).  `look_at_pose` +
# `intrinsics` above are one corner of it: up = world z (no roll: pose[2, 0] == 0), fx = fy, principal point at the centre.
POSE_KINDS = ("look_at", "roll", "target", "so3")
K_KINDS = ("centred", "free")


def _roll(pose: torch.Tensor, angle: float) -> torch.Tensor:
    """``pose`` with its camera rotated by ``angle`` about its own view axis (the right / down columns turn, forward stays)."""
    c, s = math.cos(angle), math.sin(angle)
    r = pose[:3, :3].double()
    out = pose.clone()
    out[:3, 0] = (c * r[:, 0] + s * r[:, 1]).float()
    out[:3, 1] = (-s * r[:, 0] + c * r[:, 1]).float()
    return out


def family_pose(gen: torch.Generator, centre: torch.Tensor, kind: str, target_radius: float = 0.6) -> torch.Tensor:
    """cam->world 4x4 of a camera at ``centre``: 'roll' = look at the origin, rolled by U(-pi, pi) about the view axis;
    'target' = look at a point drawn uniformly from the ball of ``target_radius`` (pitch / yaw off the centre), rolled;
    'so3' = a uniformly random orientation (Gaussian matrix -> QR, det +1): most of the grid is outside the view."""
    if kind == "look_at":
        return look_at_pose(centre)
    if kind == "so3":
        q, r = torch.linalg.qr(torch.randn(3, 3, generator=gen, dtype=torch.float64))
        q = q * torch.sign(torch.diagonal(r))[None]
        if torch.linalg.det(q) < 0:
            q[:, 2] = -q[:, 2]
        pose = torch.eye(4, dtype=torch.float64)
        pose[:3, :3] = q
        pose[:3, 3] = centre.double()
        return pose.float()
    angle = float((torch.rand((), generator=gen, dtype=torch.float64) * 2 - 1) * math.pi)
    if kind == "roll":
        return _roll(look_at_pose(centre), angle)
    if kind == "target":
        t = torch.randn(3, generator=gen, dtype=torch.float32)
        t = t / t.norm() * target_radius * float(torch.rand((), generator=gen)) ** (1.0 / 3.0)
        pose = look_at_pose(centre - t)  # the view direction of a camera at `centre` looking at t
        pose[:3, 3] = centre
        return _roll(pose, angle)
    raise ValueError(kind)


def family_intrinsics(gen: torch.Generator, width: int, height: int) -> torch.Tensor:
    """fx, fy = 0.9 W x U(0.8, 1.3) each (fx != fy), principal point up to 20 % of the image off its centre."""
    u = torch.rand(4, generator=gen, dtype=torch.float32)
    k = torch.eye(3, dtype=torch.float32)
    k[0, 0] = 0.9 * width * (0.8 + 0.5 * float(u[0]))
    k[1, 1] = 0.9 * width * (0.8 + 0.5 * float(u[1]))
    k[0, 2] = width * (0.5 + 0.4 * (float(u[2]) - 0.5))
    k[1, 2] = height * (0.5 + 0.4 * (float(u[3]) - 0.5))
    return k


def _analytic_depth(pose: torch.Tensor, k: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """Depth (camera z) of a sphere r=0.9 m inside an axis-aligned box of half-size 1.2 m.  Works on
    whatever device ``pose`` lives on (the benchmark generates its frames on the GPU)."""
    return _analytic_scene(pose, k, width, height)[0]


# Panoptic class ids the analytic scene's surfaces carry (COCO panoptic order, kmax.constants: 56 = chair's thing id is not
# needed here -- any ids in [0, 133) work; the names come from whatever class list the caller hands to discover_objects):
# the sphere, then the box's faces -x +x -y +y -z +z.  Two opposite walls share a class: two objects of one label.
SCENE_SURFACE_CLASSES = (56, 131, 131, 119, 119, 122, 100)


def _analytic_scene(pose: torch.Tensor, k: torch.Tensor, width: int, height: int, half=1.2):
    """(depth [H,W] f32, surface [H,W] int64): depth as ``_analytic_depth``; surface 0 = the sphere, 1..6 = the box
    face the ray leaves through (-x +x -y +y -z +z).  ``half``: the box's half-size, a number or one per axis."""
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
    half = torch.as_tensor(half, dtype=torch.float64, device=dev)
    inv = 1.0 / d
    t1 = (-half - o) * inv
    t2 = (half - o) * inv
    far = torch.maximum(t1, t2)
    s_box, axis = far.min(dim=-1)
    s = torch.minimum(s_sphere, s_box)
    exits_high = torch.gather(t2 >= t1, -1, axis[..., None])[..., 0]
    surface = torch.where(s_sphere <= s_box, torch.zeros_like(axis), 1 + 2 * axis + exits_high.long())
    return s.float(), surface


def blob_label_map(gen, width: int, height: int, n_blobs: int = 40, n_label_classes: int = 134, device=None):
    """A piecewise-constant class map [H,W] int64 like a panoptic segmentation's: the Voronoi cells of ``n_blobs``
    seeded points, each with a random class (SURVEY 8d draws the label map iid per pixel -- the worst case for the
    label histogram and nothing a segmentation network emits)."""
    dev = device if device is not None else (gen.device if hasattr(gen, "device") else "cpu")
    pts = torch.rand((n_blobs, 2), generator=gen, device=dev) * torch.tensor([height, width], dtype=torch.float32, device=dev)
    cls = torch.randint(0, n_label_classes, (n_blobs,), generator=gen, device=dev)
    vv, uu = torch.meshgrid(torch.arange(height, dtype=torch.float32, device=dev),
                            torch.arange(width, dtype=torch.float32, device=dev), indexing="ij")
    d2 = (vv[..., None] - pts[:, 0]) ** 2 + (uu[..., None] - pts[:, 1]) ** 2
    return cls[d2.argmin(dim=-1)]


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
    label_kind: str = "iid",
    box_half=1.2,
    pose_kind: str = "look_at",
    k_kind: str = "centred",
):
    """One frame: dict of CPU f32 tensors shaped like a B=1 batch of the reference's loaders
    (clipfusion.py:190) plus the per-frame feature map and label map that stand in for the
    CLIP / kMaX backbones."""
    c = torch.randn(3, generator=gen, dtype=torch.float32)
    c = c / c.norm() * radius
    # (the defaults draw nothing more from `gen` than they always did: the committed goldens regenerate byte for byte)
    pose = look_at_pose(c) if pose_kind == "look_at" else family_pose(gen, c, pose_kind)
    k = intrinsics(width, height) if k_kind == "centred" else family_intrinsics(gen, width, height)
    surface = None
    if depth_kind == "A":
        depth = torch.rand(height, width, generator=gen, dtype=torch.float32) * 2.0 + 1.5
    elif depth_kind == "B":
        depth, surface = _analytic_scene(pose, k, width, height, box_half)
        if pose_kind != "look_at":  # a ray that leaves the room backwards (the camera looks away from it): no depth
            depth = torch.where(torch.isfinite(depth) & (depth > 0), depth, torch.zeros_like(depth))
    else:
        raise ValueError(depth_kind)
    if missing_depth_frac > 0:
        hole = torch.rand(height, width, generator=gen) < missing_depth_frac
        depth = depth.masked_fill(hole, 0.0)
    rgb = torch.rand(height, width, 3, generator=gen, dtype=torch.float32)
    feat = torch.randn(feat_dim, npy, npx, generator=gen, dtype=torch.float32)
    if label_kind == "iid":
        labels = torch.randint(0, n_label_classes, (height, width), generator=gen, dtype=torch.int64)
    elif label_kind == "blobs":  # piecewise constant in the image
        labels = blob_label_map(gen, width, height, n_label_classes=n_label_classes)
    elif label_kind == "scene":  # the class of the surface the pixel sees: coherent in 3-D (depth B only)
        if surface is None:
            raise ValueError("label_kind='scene' needs depth_kind='B'")
        labels = torch.tensor(SCENE_SURFACE_CLASSES, dtype=torch.int64)[surface]
    else:
        raise ValueError(label_kind)
    if surface is not None and label_kind == "scene":
        return {"depth": depth[None], "rgb": rgb[None], "pose": pose[None], "K": k[None], "feat": feat[None], "labels": labels,
                "surface": surface}
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


def make_family_frames(seed: int, n_frames: int, width: int, height: int, feat_dim: int, npy: int, npx: int,
                       missing_depth_frac: float = 0.08, label_kind: str = "iid"):
    """``n_frames`` frames of the camera family real scans have (`family_pose` / `family_intrinsics`): rolled look-at, look-at
    an off-centre target, uniformly random orientations; fx != fy, principal point off the centre; cameras outside, at the
    edge of and inside a 2.56 m grid; random and analytic depth, some of it missing.  One seeded schedule, shared by the
    golden generator (the reference runs it) and the tests."""
    gen = torch.Generator().manual_seed(seed)
    pose_kinds = ("roll", "target", "target", "so3", "roll", "target")
    radii = (2.5, 2.5, 1.4, 0.6, 2.0, 0.5, 2.5)
    depth_kinds = ("A", "B", "A", "B", "B")
    frames = []
    for i in range(n_frames):
        frames.append(make_frame(gen, width, height, feat_dim, npy, npx, depth_kind=depth_kinds[i % 5], radius=radii[i % 7],
                                 missing_depth_frac=missing_depth_frac if i % 4 == 1 else 0.0, label_kind=label_kind,
                                 pose_kind=pose_kinds[i % 6], k_kind="free"))
    return frames


def feature_map_shape(width: int, height: int):
    """Patch size H/3, stride H/6 (clipfusion.py:1199-1201 uses 160/80 at 640x480)."""
    p, s = height // 3, height // 6
    return (height - p) // s + 1, (width - p) // s + 1


# ---- a whole synthetic scan for the scene-level flow (spatially_aware_ai_amd/scene.py, bench.py --scene, tests/test_scene_pipeline.py)
SCENE_CLASS_NAMES = {56: "chair", 131: "wall", 119: "floor", 122: "ceiling", 100: "window"}
REFERENCE_GRID_BOX_HALF = (1.21, 0.98, 1.10)  # with 2 cm voxels and trunc = 3 voxels the scene's bounds come out as the
                                              # reference's largest recorded grid, 127 x 104 x 116 (voxel_grid_compare.md:1-23)


def class_embeddings(dim: int, n_classes: int = 134, seed: int = 1234) -> torch.Tensor:
    """One seeded unit vector per panoptic class: what stands in for a CLIP embedding of the class in the scene flow."""
    e = torch.randn(n_classes, dim, generator=torch.Generator().manual_seed(seed))
    return e / e.norm(dim=-1, keepdim=True)


class SyntheticScan(torch.utils.data.Dataset):
    """A scan of the analytic sphere-in-a-box scene in the shape of the reference's loaders: ``imwidth`` / ``imheight`` and
    ``__getitem__ -> (rgb[H,W,3], depth[H,W], pose[4,4], K[3,3], idx)`` (clipfusion.py:190).  Per frame it also holds what the
    backbones would say: a panoptic map (the class of the surface every pixel sees) and a CLIP-shaped feature map whose
    cell (i, j) is the embedding of the class at the cell's centre plus noise -- so that objects, segment colours and a text
    query over the reconstruction mean something."""

    def __init__(self, seed, n_frames, width, height, feat_dim, box_half=1.2, noise=0.3):
        self.imwidth, self.imheight = width, height
        self.npy, self.npx = feature_map_shape(width, height)
        self.patch, self.stride = height // 3, height // 6
        self.emb = class_embeddings(feat_dim)
        gen = torch.Generator().manual_seed(seed)
        self.frames = []
        cls_of = torch.tensor(SCENE_SURFACE_CLASSES, dtype=torch.int64)
        cy = torch.arange(self.npy) * self.stride + self.patch // 2
        cx = torch.arange(self.npx) * self.stride + self.patch // 2
        for _ in range(n_frames):
            f = make_frame(gen, width, height, feat_dim, self.npy, self.npx, depth_kind="B", label_kind="scene", box_half=box_half)
            centre_cls = cls_of[f["surface"][cy][:, cx]]  # [npy, npx]
            f["feat"] = (self.emb[centre_cls].permute(2, 0, 1) + noise * f["feat"][0])[None].contiguous()
            self.frames.append(f)

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        f = self.frames[i]
        return f["rgb"][0], f["depth"][0], f["pose"][0], f["K"][0], i


class ReplayClip:
    """Backbone stand-in for a ``SyntheticScan``: ``img_inference_tiled`` hands back the scan's feature maps in call order
    (one ``integrate`` per frame, as the reference drives it); the text side maps a class name to its embedding."""

    def __init__(self, scan, device="cuda", class_names=None):
        self.feature_dim = scan.emb.shape[1]
        self.maps = [f["feat"].to(device) for f in scan.frames]
        self.emb = scan.emb
        self.names = class_names
        self.calls = 0

    def img_inference_tiled(self, rgb, patch_size, patch_stride):
        m = self.maps[self.calls % len(self.maps)]
        self.calls += 1
        return m

    def encode_text_with_prompt_ensemble(self, texts, device, prompt_templates=None):
        rows = []
        for t in texts:
            if self.names is not None and t in self.names:
                rows.append(self.emb[self.names.index(t)])
            else:  # not a class of the scene: a seeded direction of its own
                v = torch.randn(self.feature_dim, generator=torch.Generator().manual_seed(zlib.crc32(t.encode())))
                rows.append(v / v.norm())
        return torch.stack(rows).to(device)


class ReplaySeg:
    def __init__(self, scan, device="cuda"):
        self.maps = [f["labels"].to(device) for f in scan.frames]
        self.calls = 0

    def run_on_image(self, rgb_chw):
        m = self.maps[self.calls % len(self.maps)]
        self.calls += 1
        return m


def scene_class_names(n_classes: int = 134):
    return [SCENE_CLASS_NAMES.get(i, f"class{i}") for i in range(n_classes)]


def scene_class_colors(n_classes: int = 134):
    g = torch.Generator().manual_seed(77)
    return torch.randint(0, 256, (n_classes, 3), generator=g).tolist()
