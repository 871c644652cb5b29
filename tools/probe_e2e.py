"""Development probe: host time per integrate() call with the deferred backbone (one frame per call)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from spatially_aware_ai_amd import ClipFusion, synthetic as syn
from spatially_aware_ai_amd.backbones import RandomViTB32
from spatially_aware_ai_amd.clipfusion import Clip
g = syn.make_grid(256)
npy, npx = syn.feature_map_shape(640, 480)
dev = torch.device("cuda", 0)
depth, rgb, poses, ks, feat = bench.gen_frames_gpu(256, 640, 480, 512, npy, npx, "A", 1000, dev)
clip = Clip("ViT-B-32 (random weights)", None, backbone=RandomViTB32(), tokenizer=None).to(dev).eval()
clip.requires_grad_(False)
for defer in (True, False):
    fz = ClipFusion(g.origin, g.voxel_size, g.nvox, g.trunc, False, clip, None, 160, 80, keep_xyz_world=False, defer_backbone=defer).to(dev)
    for rep in range(2):
        fz.reset(); ts = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for i in range(256):
                t = time.perf_counter()
                fz.integrate(depth[i:i+1], rgb[i:i+1], poses[i:i+1], ks[i:i+1])
                ts.append(time.perf_counter() - t)
            t = time.perf_counter(); fz.flush(); ts.append(time.perf_counter() - t)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"defer_backbone={defer} rep {rep}: {256/dt:.0f} frames/s; host us/call buckets of 32:",
              " ".join(f"{sum(ts[j:j+32])/32*1e6:.0f}" for j in range(0, 256, 32)), "flush ms", round(ts[-1]*1e3, 1),
              "slowest", sorted(((round(x*1e3, 1), j) for j, x in enumerate(ts)), reverse=True)[:4])
