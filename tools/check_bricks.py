"""Development check (GPU): the brick form of the window row kernel against the frame-ordered form on the same frames.
Everything but clip_feat must be equal; clip_feat within rounding.  usage: python tools/check_bricks.py [quick]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from spatially_aware_ai_amd import ClipFusion, ClipSeemFusion, _abi
from spatially_aware_ai_amd import synthetic as syn


class FakeClip:
    def __init__(self, dim):
        self.feature_dim = dim


class FakeSeg:
    pass


def run(form, nvox, dim, seem, accum, n_frames, fdt, kind, seed):
    os.environ["SAF_WIN_FORM"] = form
    w, h = 64, 48
    npy, npx = syn.feature_map_shape(w, h)
    grid = syn.make_grid(nvox, side=2.56 * nvox[0] / max(nvox))
    frames = syn.make_frames(seed, n_frames, width=w, height=h, feat_dim=dim, npy=npy, npx=npx, depth_kind=kind, missing_depth_frac=0.05)
    if n_frames > 40:
        for i in range(12, min(112, n_frames)):
            frames[i] = dict(frames[i], depth=frames[11]["depth"], pose=frames[11]["pose"], K=frames[11]["K"])
    clip, seg = FakeClip(dim), FakeSeg()
    if seem:
        fz = ClipSeemFusion(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, False, 10, 10, clip, seg, keep_xyz_world=False, feat_dtype=fdt).cuda()
    else:
        fz = ClipFusion(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, False, clip, None, 10, 10, keep_xyz_world=False, feat_dtype=fdt).cuda()
    fz.accum_mode = accum
    cat = lambda k: torch.cat([f[k] for f in frames]).cuda()
    labs = [f["labels"].float().cuda() for f in frames] if seem else None
    for rep in range(2):  # twice: the second pass reads rows the first one wrote
        fz.integrate_features(cat("depth"), cat("rgb"), cat("pose"), cat("K"), cat("feat"), labs)
    torch.cuda.synchronize()
    return fz


def main():
    cases = [
        ((33, 30, 41), 512, True, _abi.SAF_RUNNING_MEAN, 40, torch.float32, "B"),
        ((33, 30, 41), 256, False, _abi.SAF_SUM, 150, torch.float32, "B"),
        ((64, 64, 64), 256, False, _abi.SAF_RUNNING_MEAN, 75, torch.float32, "A"),
        ((32, 16, 128), 512, True, _abi.SAF_RUNNING_MEAN, 17, torch.float32, "A"),
        ((127, 104, 116), 512, True, _abi.SAF_RUNNING_MEAN, 130, torch.float32, "B"),
        ((33, 30, 41), 512, True, _abi.SAF_RUNNING_MEAN, 36, torch.bfloat16, "B"),
        ((64, 48, 64), 1024, False, _abi.SAF_SUM, 17, torch.bfloat16, "A"),
    ]
    bad = 0
    for i, (nvox, dim, seem, accum, nf, fdt, kind) in enumerate(cases):
        t0 = time.time()
        a = run("rows", nvox, dim, seem, accum, nf, fdt, kind, 900 + i)
        b = run("bricks", nvox, dim, seem, accum, nf, fdt, kind, 900 + i)
        msg = []
        for name in ("weight", "tsdf_weight", "tsdf", "rgb") + (("labels_one_hot",) if seem else ()):
            if not torch.equal(getattr(a, name), getattr(b, name)):
                msg.append(f"{name} DIFFERS ({int((getattr(a, name) != getattr(b, name)).sum())} elements)")
                bad += 1
        fa, fb = a.clip_feat.float(), b.clip_feat.float()
        scale = fa.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
        rel = ((fa - fb).abs() / scale).max().item()
        tol = 4e-2 if fdt == torch.bfloat16 else 5e-6
        if not (rel <= tol):
            msg.append(f"clip_feat rel-to-row-max err {rel:.3g} > {tol}")
            bad += 1
        sa, sb = a.stats(), b.stats()
        if sa != sb:
            msg.append(f"stats differ {sa} {sb}")
            bad += 1
        print(f"case {i} nvox {nvox} D {dim} seem {seem} accum {accum} frames {nf} {fdt} depth {kind}: feat err {rel:.3g} "
              f"rows {sb['window_rows']} valid {sb['valid']} {'OK' if not msg else msg}  ({time.time() - t0:.1f} s)", flush=True)
    print("FAILED" if bad else "ALL OK", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
