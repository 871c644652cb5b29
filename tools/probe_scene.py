"""Development probe (GPU box): where the scene flow's per-frame host time goes -- the loader alone, the host -> device copies,
integrate() with resident frames, backproject_pcd."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatially_aware_ai_amd import synthetic as syn  # noqa: E402
from spatially_aware_ai_amd.clip_seem_fusion import ClipSeemFusion  # noqa: E402
from spatially_aware_ai_amd.clipfusion import backproject_pcd, scene_bounds  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
names = syn.scene_class_names()
scan = syn.SyntheticScan(4, n, 640, 480, 512, box_half=syn.REFERENCE_GRID_BOX_HALF)


def timed(what, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"{what:60s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms per frame", flush=True)
    return r


def loader_only(pin=False):
    for _ in torch.utils.data.DataLoader(scan, batch_size=1, num_workers=0, pin_memory=pin):
        pass


def loader_to_device(pin=False):
    for rgb, depth, pose, K, _ in torch.utils.data.DataLoader(scan, batch_size=1, num_workers=0, pin_memory=pin):
        depth.to(dev, non_blocking=pin), rgb.to(dev, non_blocking=pin), pose.to(dev, non_blocking=pin), K.to(dev, non_blocking=pin)


timed("DataLoader(batch_size=1) alone", loader_only)
timed("DataLoader + 4 x .to(device)", loader_to_device)
timed("DataLoader(pin_memory) + 4 x .to(device, non_blocking)", lambda: loader_to_device(True))
timed("plain indexing + .to(device)", lambda: [[t.to(dev) for t in scan[i][:4]] for i in range(n)])
xyz = timed("backproject_pcd", lambda: backproject_pcd(scan, batch_size=1, max_depth=4)[0])
origin, nvox = scene_bounds(xyz, 0.02, 0.06)
clip, seg = syn.ReplayClip(scan, dev, names), syn.ReplaySeg(scan, dev)
fz = ClipSeemFusion(origin, 0.02, nvox, 0.06, False, 160, 80, clip, seg).to(dev)
res = [[t[None].to(dev) for t in scan[i][:4]] for i in range(n)]


def integ():
    for rgb, depth, pose, K in res:
        fz.integrate(depth, rgb, pose, K)
    fz.flush()


timed("integrate() with resident frames (first pass)", integ)
fz.reset()
timed("integrate() with resident frames", integ)
fz.reset()


def full():
    for rgb, depth, pose, K, _ in torch.utils.data.DataLoader(scan, batch_size=1, num_workers=0):
        fz.integrate(depth.to(dev), rgb.to(dev), pose.to(dev), K.to(dev))
    fz.flush()


timed("loader + .to + integrate (the reference's loop)", full)
