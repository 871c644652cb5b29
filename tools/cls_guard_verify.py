#!/usr/bin/env python3
"""Development: the classification's guarded pixel path against the reference's chain, voxel slot by voxel slot (SAF_CLS_VERIFY=1 in the
environment: both are computed, disagreements counted in stats[7]).  Runs bench-like jobs at several image sizes,
both depth distributions; prints tests and disagreements.  tools/cls_guard_verify.sh builds, runs, rebuilds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from spatially_aware_ai_amd import _abi
from spatially_aware_ai_amd import synthetic as syn
from test_brick_form import _build, _frames, _fuse

total = 0
for (nvox, w, h, n, kind, seed) in [((128, 128, 128), 640, 480, 256, "A", 1), ((128, 128, 128), 640, 480, 256, "B", 2),
                                    ((96, 128, 64), 320, 240, 128, "A", 3), ((64, 96, 128), 1280, 960, 64, "B", 4),
                                    ((127, 104, 116), 333, 517, 128, "A", 5), ((128, 128, 128), 1920, 1080, 48, "A", 6)]:
    grid = syn.make_grid(nvox, side=2.56)
    frames = _frames(seed, n, 256, kind, w=w, h=h)
    fz = _fuse(_build(grid, 256, False, _abi.SAF_RUNNING_MEAN, torch.float32), frames, False)
    s = fz.fuse_stats.cpu().tolist()
    print(f"grid {nvox} image {w}x{h} depth {kind}: {n} frames, tsdf-valid {s[1]}, valid {s[0]}, window rows {s[5]}, DISAGREEMENTS {s[7]}")
    total += s[7]
print("total disagreements:", total)
sys.exit(1 if total else 0)
