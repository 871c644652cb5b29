"""Development probe: how long does saf_clear_unwritten_rows take, by fraction of unwritten rows?"""
import ctypes as C, sys, time, torch
sys.path.insert(0, ".")
from spatially_aware_ai_amd import ClipFusion, synthetic as syn
from spatially_aware_ai_amd._lib import lib, check
class R: feature_dim = 512
g = syn.make_grid(256)
fz = ClipFusion(g.origin, g.voxel_size, g.nvox, g.trunc, False, R(), None, 160, 80, keep_xyz_world=False).cuda()
n = fz._buffers["weight"].numel()
for frac in (0.0, 0.01, 0.1, 0.5, 1.0):
    w = (torch.rand(n, device="cuda") >= frac).int()
    fz._buffers["weight"].copy_(w)
    vol = fz._c_volume(for_fuse=True)
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter()
        check(lib().saf_clear_unwritten_rows(C.byref(vol), 0, n, torch.cuda.current_stream().cuda_stream), "clear")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"unwritten {frac:5.2f}: {dt*1e3:8.3f} ms  ({frac*n*2048/dt/1e12:.2f} TB/s of zero writes)")
