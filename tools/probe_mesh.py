"""Development probe: where extract_mesh() spends its time (256^3 x 512, 64 frames of scene B)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from spatially_aware_ai_amd import ClipFusion, synthetic as syn
from spatially_aware_ai_amd import clipfusion as cf
class R: feature_dim = 512
dev = torch.device("cuda", 0)
g = syn.make_grid(256)
npy, npx = syn.feature_map_shape(640, 480)
depth, rgb, poses, ks, feat = bench.gen_frames_gpu(64, 640, 480, 512, npy, npx, "B", 1000, dev)
fz = ClipFusion(g.origin, g.voxel_size, g.nvox, g.trunc, False, R(), None, 160, 80, keep_xyz_world=False).to(dev)
fz.integrate_features(depth, rgb, poses, ks, feat); fz.flush()
fz.extract_mesh(); torch.cuda.synchronize()
def T(f, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for rep in range(2):
    (vf), t_mc = T(fz._marching_cubes_vertices)
    verts, faces = vf
    (_), t_s = T(fz.sample_mesh_vertices, verts)
    (_), t_w = T(fz._verts_world, verts)
    (_), t_all = T(fz.extract_mesh)
    print(f"marching cubes {t_mc:.2f} ms, sampling {t_s:.2f} ms, verts_world {t_w:.2f} ms; extract_mesh {t_all:.2f} ms; verts type {type(verts).__name__} faces type {type(faces).__name__}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); fz.extract_mesh(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
