"""Development probe: where does the host time of the one-frame-per-call loop go?  (cProfile of one 512-frame job)"""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
import bench
from spatially_aware_ai_amd import ClipFusion, synthetic as syn
class R: feature_dim = 512
g = syn.make_grid(256)
npy, npx = syn.feature_map_shape(640, 480)
dev = torch.device("cuda", 0)
depth, rgb, poses, ks, feat = bench.gen_frames_gpu(512, 640, 480, 512, npy, npx, "A", 1000, dev)
fz = ClipFusion(g.origin, g.voxel_size, g.nvox, g.trunc, False, R(), None, 160, 80, keep_xyz_world=False).to(dev)
def job():
    fz.reset()
    for i in range(512):
        fz.integrate_features(depth[i:i+1], rgb[i:i+1], poses[i:i+1], ks[i:i+1], feat[i:i+1])
    fz.flush()
job(); torch.cuda.synchronize()
for ring in (64, 128, 256):
    fz._QUEUE_FRAMES = ring
    fz.__dict__["_stage"] = None
    job(); torch.cuda.synchronize()
    t0 = time.perf_counter(); job(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"ring {ring}: host {th*1e3:.1f} ms, total {tt*1e3:.1f} ms -> {512/tt:.0f} frames/s")
fz._QUEUE_FRAMES = 256
pr = cProfile.Profile(); pr.enable(); job(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
# bulk calls first (as bench.py does), then the loop again
for _ in range(3):
    fz.reset(); fz.integrate_features(depth, rgb, poses, ks, feat); fz.flush()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); job(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"after bulk, rep {rep}: host {th*1e3:.1f} ms, total {tt*1e3:.1f} ms -> {512/tt:.0f} frames/s")
import ctypes as C
from spatially_aware_ai_amd._lib import lib
p = lib().saf_profiler_create(8000)
for rep in range(2):
    t0 = time.perf_counter(); job(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"with 16000 live events, rep {rep}: host {th*1e3:.1f} ms, total {tt*1e3:.1f} ms -> {512/tt:.0f} frames/s")
lib().saf_profiler_destroy(p)
t0 = time.perf_counter(); job(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
print(f"events destroyed: host {th*1e3:.1f} ms, total {tt*1e3:.1f} ms -> {512/tt:.0f} frames/s")
import os
os.environ["SAF_X"] = "0"
t0 = time.perf_counter(); job(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
print(f"after os.environ set: host {th*1e3:.1f} ms, total {tt*1e3:.1f} ms -> {512/tt:.0f} frames/s")
del os.environ["SAF_X"]
t0 = time.perf_counter(); job(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
print(f"after os.environ del: host {th*1e3:.1f} ms, total {tt*1e3:.1f} ms -> {512/tt:.0f} frames/s")
t0 = time.perf_counter()
for _ in range(1000): os.environ.get("PYTORCH_NVML_BASED_CUDA_CHECK")
print("os.environ.get x1000:", (time.perf_counter() - t0) * 1e3, "ms")
t0 = time.perf_counter()
for _ in range(1000): torch.cuda.is_available()
print("torch.cuda.is_available x1000:", (time.perf_counter() - t0) * 1e3, "ms")
