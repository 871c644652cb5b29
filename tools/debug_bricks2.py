import os, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from spatially_aware_ai_amd import _abi, synthetic as syn
import test_brick_form as T

def run(nvox, dim, seem, nf, kind, rest, fdt=torch.float32):
    grid = syn.make_grid(nvox, side=2.56 * nvox[0] / max(nvox))
    frames = T._frames(7000 + dim + nf, nf, dim, kind, rest=rest)
    os.environ.pop("SAF_WIN_FORM", None)
    one = T._fuse(T._build(grid, dim, seem, 0, fdt, defer=False), frames, seem, per_call=7)
    os.environ["SAF_WIN_FORM"] = "bricks"
    win = T._fuse(T._build(grid, dim, seem, 0, fdt), frames, seem)
    a, b = one.clip_feat.float().cpu(), win.clip_feat.float().cpu()
    scale = a.abs().amax(1, keepdim=True).clamp_min(1e-30)
    err = ((a - b).abs() / scale)
    bad = torch.nonzero(err.amax(1) > 1e-4).flatten()
    print(nvox, dim, seem, nf, kind, rest, "max err", float(err.max()), "bad rows", bad.numel(), "of", int((a.abs().amax(1) > 0).sum()))
    nx, ny, nz = nvox
    w = one.weight.cpu()
    for r in bad[:5].tolist():
        x, y, z = r // (ny * nz), (r // nz) % ny, r % nz
        ch = torch.nonzero(err[r] > 1e-4).flatten()
        print("   voxel", (x, y, z), "local", (x % 4, y % 4, z % 4), "weight", int(w[r]), "bad channels", ch.numel(), ch[:6].tolist(), ch[-3:].tolist(),
              "a", a[r, ch[:3]].numpy().round(4), "b", b[r, ch[:3]].numpy().round(4))

for env in ({}, {"SAF_BRICK_SPLIT": "0"}, {"SAF_WIN_OVERLAP": "0"}, {"SAF_BRICK_POOL_CAP": "0"}):
    os.environ.update(env)
    print(env)
    run((31, 26, 29), 256, False, 130, "A", None)
    run((33, 30, 41), 256, False, 130, "B", None)
    for k in env: os.environ.pop(k)
