import sys, time, torch
sys.path.insert(0, ".")
import bench
from spatially_aware_ai_amd import ClipFusion, synthetic as syn
import spatially_aware_ai_amd.clipfusion as cf
class R: feature_dim = 512
g = syn.make_grid(256)
npy, npx = syn.feature_map_shape(640, 480)
dev = torch.device("cuda", 0)
depth, rgb, poses, ks, feat = bench.gen_frames_gpu(512, 640, 480, 512, npy, npx, "A", 1000, dev)
fz = ClipFusion(g.origin, g.voxel_size, g.nvox, g.trunc, False, R(), None, 160, 80, keep_xyz_world=False).to(dev)
def timed(name):
    orig = getattr(cf._FusionVolumeMixin, name)
    def w(self, *a, **k):
        t = time.perf_counter(); r = orig(self, *a, **k); dt = time.perf_counter() - t
        if dt > 1e-3: print(f"   {name}: {dt*1e3:.2f} ms")
        return r
    setattr(cf._FusionVolumeMixin, name, w)
for n in ("_flush_pending", "_fuse_now", "_make_frames", "_sync_volume", "_get_workspace", "_c_volume"):
    timed(n)
L = cf.lib()
class LW:
    def __getattr__(self, k):
        f = getattr(L, k)
        def w(*a):
            t = time.perf_counter(); r = f(*a); dt = time.perf_counter() - t
            if dt > 1e-3: print(f"   C {k}: {dt*1e3:.2f} ms")
            return r
        return w
cf.lib = lambda: LW()
def job():
    fz.reset()
    for i in range(512):
        fz.integrate_features(depth[i:i+1], rgb[i:i+1], poses[i:i+1], ks[i:i+1], feat[i:i+1])
    fz.flush()
for rep in range(3):
    print("job", rep)
    t0 = time.perf_counter(); job(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f" host {th*1e3:.1f} ms, total {tt*1e3:.1f} ms -> {512/tt:.0f} frames/s")
print("after bulk calls")
for _ in range(2):
    fz.reset(); fz.integrate_features(depth, rgb, poses, ks, feat); fz.flush()
torch.cuda.synchronize()
for rep in range(2):
    print("job", rep)
    t0 = time.perf_counter(); job(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f" host {th*1e3:.1f} ms, total {tt*1e3:.1f} ms -> {512/tt:.0f} frames/s")
