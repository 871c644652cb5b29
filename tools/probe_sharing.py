"""Development probe (CPU or GPU): how many hits of a 128-frame window share (frame, map cell) inside a box of voxels -- the
factor by which a box-resident row kernel cuts its tap gathers -- and how many hits / rows / distinct map rows a box has.
Statistical (plain float math, not the kernels' exact rounding).  usage: probe_sharing.py [A|B] [n_boxes]"""
import sys, torch
sys.path.insert(0, ".")
import bench
from spatially_aware_ai_amd import synthetic as syn
dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
g = syn.make_grid(256)
npy, npx = syn.feature_map_shape(640, 480)
kind = sys.argv[1] if len(sys.argv) > 1 else "A"
nbox = int(sys.argv[2]) if len(sys.argv) > 2 else 48
depth, rgb, poses, ks, feat = bench.gen_frames_gpu(128, 640, 480, 8, npy, npx, kind, 1000, dev)
W, H = 640, 480
origin = torch.as_tensor(g.origin, dtype=torch.float32, device=dev); vs = float(g.voxel_size); trunc = float(g.trunc)
gen = torch.Generator(device="cpu").manual_seed(0)
shapes = {"1x1x64": (1, 1, 64), "4x4x4": (4, 4, 4), "4x4x8": (4, 4, 8), "4x8x4": (4, 8, 4), "4x4x16": (4, 4, 16), "8x8x4": (8, 8, 4), "4x8x8": (4, 8, 8),
          "8x8x8": (8, 8, 8), "4x4x64": (4, 4, 64)}
for name, (bx, by, bz) in shapes.items():
    hits = groups = rows = taps = 0
    maxh = 0
    for _ in range(nbox):
        x0 = int(torch.randint(0, 256 // bx, (1,), generator=gen)) * bx
        y0 = int(torch.randint(0, 256 // by, (1,), generator=gen)) * by
        z0 = int(torch.randint(0, 256 // bz, (1,), generator=gen)) * bz
        ix, iy, iz = torch.meshgrid(torch.arange(x0, x0 + bx), torch.arange(y0, y0 + by), torch.arange(z0, z0 + bz), indexing="ij")
        X = (torch.stack([ix, iy, iz], -1).reshape(-1, 3).float().to(dev) * vs + origin)  # [V,3]
        touched = torch.zeros(X.shape[0], dtype=torch.bool, device=dev)
        bh = 0
        for f in range(128):
            R, t = poses[f, :3, :3], poses[f, :3, 3]
            cam = (X - t) @ R  # R^T (X - t)
            uvz = cam @ ks[f].T
            z = uvz[:, 2]
            u, v = uvz[:, 0] / z, uvz[:, 1] / z
            px, py = torch.round(u).long(), torch.round(v).long()
            inb = (z > 0) & (px >= 0) & (px < W) & (py >= 0) & (py < H)
            d = torch.zeros_like(z)
            d[inb] = depth[f][py[inb], px[inb]]
            sdf = (d - z) / trunc
            valid = inb & (sdf.abs() <= 1)
            n = int(valid.sum())
            if n == 0:
                continue
            gx, gy = (u[valid] + 0.5) / W * npx - 0.5, (v[valid] + 0.5) / H * npy - 0.5
            cy, cx = torch.floor(gy).long().clamp(-2, npy), torch.floor(gx).long().clamp(-2, npx)
            cell = cy * 64 + cx
            hits += n
            bh += n
            groups += int(torch.unique(cell).numel())
            tp = torch.cat([cell, cell + 1, cell + 64, cell + 65])
            taps += int(torch.unique(tp).numel())
            touched |= valid
        rows += int(touched.sum())
        maxh = max(maxh, bh)
    print(f"depth {kind} box {name:8s}: rows/box {rows/nbox:7.1f} hits/box {hits/nbox:8.1f} (max {maxh}) hits/row {hits/max(1,rows):5.2f}  "
          f"hits per (frame, cell) group {hits/max(1,groups):6.2f}  distinct map rows/box {taps/nbox:7.1f} = {hits*4/max(1,taps):5.2f} hit-taps per map row", flush=True)
