#!/usr/bin/env python3
"""Benchmark of the fused hot path on MI355X: RGB-D frames/s fused into a voxel feature volume.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.

Workload (BASELINE.json metric / configs[3]): per rank 512 synthetic 640x480 RGB-D frames with
random poses (depth distribution A) fused into a per-rank 256^3 x 512 fp32 grid;
ranks shard frames and merge their grids with one RCCL reduction at the end of the job.

A *step* is one whole job per rank: reset the volume, fuse this rank's frames in ONE C-ABI call with no host sync (the
windowed path: per window of 128 frames four classification launches and one row kernel, the next window classified beside
the row kernel), finish the deferred clear, and -- for N > 1 -- the single merge (reduce-scatter of the per-rank SUM volumes
+ local divide).  Inputs are resident in HBM before the timed region.  value = N * frames_per_rank * K / max-over-ranks
wall time of the K steps.

N > 1: when WORLD_SIZE is not set, `python bench.py --gpus N` starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before this process touches the GPU), forwards
rank 0's JSON line and exits non-zero if any rank fails.  An N > 1 run times BASELINE config 4's job (frames sharded, one
per-rank volume, one RCCL merge per job) in two forms over the same K steps -- merged after the fusion, nothing overlapped
(`serial_merge`), and merged slab by slab behind its own fusion (`slab_pipelined_merge`; distributed.fuse_merge_pipelined) --
proves both (weight sums against the kernels' valid counts) and reports the faster one as `value` (`headline_region` says
which); the same K steps are also timed with each job's merge overlapped with the NEXT job's fusion (`overlapped_merge`: a
stream of jobs, not config 4) and with the VOXELS sharded instead of the frames (`voxel_sharded`: balanced slabs, the
frames all-gathered in segments beside the fusion, no merge).  The merge's collectives are probed on a small tensor first (a
collective that raised inside a volume is never retried).  After the timed regions an untimed integrity pass proves the merge: weight sums against the kernels' valid counts,
and the merged voxel shard of a small sharded job against a single-rank fusion of all its frames (`merge_check`).

Other passes: --query (BASELINE config 5: the text-query scans), --api-b1 N (one frame per integrate() call through the
deferred window queue), --end-to-end N (a ViT-B/32-shaped backbone in front; 128 frames by default), --depth-kind B, --labels,
--feat-dtype bf16, --grid nx,ny,nz.  The default single-GPU run also reports `side_workloads` (BASELINE configs 2, 3 and 5, the
coherent scene, config 3 end to end: about 40 s; --no-side skips them), `hbm_copy_GBps` (this box's copy rate, measured in the
run) and `slab_by_slab_fuse` (the job fused into 8 x-slabs one after the other against one call).

Extra objects on the JSON line:
  roofline     : the dominant kernel (fuse_window_kernel): algorithmic bytes per launch (from the row / voxel counters
                 the kernels emit, SURVEY.md §8d formula) / its average launch duration, measured with HIP events on the
                 launch stream inside the timed region; `isolated`, `warm_volume`, `frame_at_a_time` beside it.
  cpu_baseline : the CPU oracle (a parity-checked port of the reference's algorithm, OpenMP over
                 the host cores of this box) on a bounded sample of the same frames, rank 0, N=1.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from spatially_aware_ai_amd import _abi  # noqa: E402
from spatially_aware_ai_amd import synthetic as syn  # noqa: E402
from spatially_aware_ai_amd._lib import check, lib  # noqa: E402

# frames per window of the windowed path (include/saf.h SAF_WINDOW_FRAMES; SAF_WIN_FRAMES=64 selects the shorter form)
WIN = 64 if os.environ.get("SAF_WIN_FRAMES") == "64" else 128
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", default="256", help="voxels per axis, or nx,ny,nz (a ragged grid like the reference's 127,104,116; "
                    "the longest edge spans 2.56 m)")
    ap.add_argument("--dim", type=int, default=512, help="feature dim D")
    ap.add_argument("--frames", type=int, default=512, help="frames per rank per step")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--depth-kind", default="A", choices=["A", "B", "Z"],
                    help="A: iid depth per pixel (the worst case); B: a coherent analytic scene; Z: no valid depth (nothing hits: the fixed cost)")
    ap.add_argument("--pose-kind", default="look_at", choices=["look_at", "free"],
                    help="look_at: SURVEY 8d's cameras (no roll, centred isotropic K); free: rolled / off-centre cameras with fx != fy "
                         "(synthetic.family_pose): what a hand-held scan's poses look like (clipfusion.py:308-312)")
    ap.add_argument("--unique-frames", type=int, default=512,
                    help="distinct synthetic frames resident per rank (cycled to --frames)")
    ap.add_argument("--merge", default="reduce_scatter", choices=["reduce_scatter", "all_reduce"])
    ap.add_argument("--no-overlap-merge", action="store_true",
                    help="N > 1: skip the second timed region (merge of job k overlapped with the fusion of job k+1)")
    ap.add_argument("--no-voxel-sharded", action="store_true",
                    help="N > 1: skip the third timed region (every rank fuses every frame into its slab of the volume)")
    ap.add_argument("--check-frames", type=int, default=4,
                    help="N > 1: frames per rank of the untimed merge-integrity job (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to rehearse the control flow)")
    ap.add_argument("--feat-dtype", default="f32", choices=["f32", "bf16"],
                    help="feature-volume dtype: f32 = the reference layout (headline); bf16 = BASELINE config 3")
    ap.add_argument("--labels", action="store_true",
                    help="ClipSeemFusion path: panoptic label histogram + bilinear rgb (BASELINE config 3)")
    ap.add_argument("--label-kind", default="iid", choices=["iid", "world"],
                    help="--labels: panoptic maps iid per pixel (SURVEY 8d) or consistent in 3-D (world_label_maps)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile-events", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes of one job after the timed "
                         "regions, about a minute); the committed profiles/ figure is quoted instead")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side workloads of the default single-GPU run (configs 2, 3, 5, the coherent scene, the HBM copy rate)")
    ap.add_argument("--end-to-end", type=int, default=-1, metavar="FRAMES",
                    help="after the timed region, also time FRAMES frames through the reference-shaped API with a "
                         "ViT-B/32-shaped random-weight CLIP image tower in front of the fuse (reported separately)")
    ap.add_argument("--e2e-batch", type=int, default=1, help="frames per integrate() call in the end-to-end pass")
    ap.add_argument("--e2e-dtype", default="bf16", choices=["f32", "bf16"], help="backbone compute dtype")
    ap.add_argument("--e2e-tile-batch", type=int, default=0, help="tiles per encode_image call (0 = the Clip class's default)")
    ap.add_argument("--api-b1", type=int, default=-1, metavar="FRAMES",
                    help="also time FRAMES frames through integrate_features() ONE FRAME PER CALL (the reference's loop, "
                         "clipfusion.py:1125-1133) with the deferred window queue behind it; reported as api_b1")
    ap.add_argument("--queries", type=int, default=1000, help="--query: number of target text queries")
    ap.add_argument("--query-wide-only", action="store_true", help="--query: skip the fp32 L=5 / L=63 cases")
    ap.add_argument("--side-traffic", metavar="JSON", default=None,
                    help="measure the side workloads' HBM counter traffic (two profiler passes per workload) into this file and exit")
    ap.add_argument("--scene", action="store_true",
                    help="the scene-level flow of the reference's manager (clip_seem_fusion.py:247-437, then :482-561) on one "
                         "synthetic scan: seconds per stage and in total, one JSON line")
    ap.add_argument("--scene-frames", type=int, default=300)
    ap.add_argument("--query", action="store_true",
                    help="benchmark the text-query scan instead (BASELINE config 5 and the reference's L = 5 / L = 63 scans)")
    ap.add_argument("--profile-stride", type=int, default=4,
                    help="record HIP events around the kernels of every n-th frame of the timed region")
    a = ap.parse_args()
    g3 = tuple(int(x) for x in str(a.grid).split(","))
    if len(g3) == 1:
        g3 = g3 * 3
    assert len(g3) == 3 and min(g3) > 0, "--grid takes N or nx,ny,nz"
    a.grid3 = g3
    if a.api_b1 < 0:  # default: on for the plain single-GPU run (the reference's call pattern, reported beside the bulk rate)
        a.api_b1 = 512 if (a.gpus == 1 and not a.query and not a.scene and not a.no_side) else 0
    if a.end_to_end < 0:  # default: on for the plain single-GPU run (128 frames, one frame per integrate() call, bf16 tower)
        a.end_to_end = 128 if (a.gpus == 1 and not a.query and not a.labels and a.dim == 512 and not a.no_side) else 0
    a.grid = g3[0] if g3[0] == g3[1] == g3[2] else "x".join(str(x) for x in g3)  # (label; cubic grids keep the integer)
    return a


def gen_frames_gpu(n, width, height, dim, npy, npx, depth_kind, seed, device, pose_kind="look_at"):
    """n frames resident on the device.  Poses/intrinsics come from the seeded CPU generator of
    synthetic.py; the bulky per-pixel data is drawn on the device (seeded) to keep start-up short.
    pose_kind "free": the camera family real scans have (synthetic.family_pose / family_intrinsics: look-at the origin or an
    off-centre target, ROLLED by U(-pi, pi) about the view axis; fx != fy, principal point up to 20 % off the centre)."""
    gen = torch.Generator().manual_seed(seed)
    poses, ks = [], []
    for i in range(n):
        c = torch.randn(3, generator=gen)
        if pose_kind == "free":
            poses.append(syn.family_pose(gen, c / c.norm() * 2.5, ("roll", "target")[i % 2]))
            ks.append(syn.family_intrinsics(gen, width, height))
            continue
        poses.append(syn.look_at_pose(c / c.norm() * 2.5))
        ks.append(syn.intrinsics(width, height))
    poses = torch.stack(poses).to(device)
    ks = torch.stack(ks).to(device)
    g = torch.Generator(device=device).manual_seed(seed)
    if depth_kind == "A":
        depth = torch.rand((n, height, width), generator=g, device=device) * 2.0 + 1.5
    elif depth_kind == "Z":
        depth = torch.zeros((n, height, width), device=device)
    else:
        depth = torch.stack([syn._analytic_depth(p, k, width, height) for p, k in zip(poses, ks)])
        if pose_kind == "free":  # (a ray that leaves the room backwards: no depth)
            depth = torch.where(torch.isfinite(depth) & (depth > 0), depth, torch.zeros_like(depth))
    rgb = torch.rand((n, height, width, 3), generator=g, device=device)
    feat = torch.randn((n, dim, npy, npx), generator=g, device=device)
    return depth, rgb, poses, ks, feat


def world_label_maps(depth, poses, ks, n_seeds=40, n_classes=134, seed=78):
    """Panoptic maps that are consistent in 3-D, as a segmentation of a static scene is: the class of a pixel is the class of the
    Voronoi cell (of `n_seeds` seeded points in the scene cube) that the surface point it sees -- un-projected with its depth --
    lies in.  Whatever frame observes a voxel's surroundings then reports the same class, which SURVEY 8d's iid map (the worst
    case for the label histogram) never does.  [F,H,W] float32 on the frames' device."""
    dev = depth.device
    g = torch.Generator(device=dev).manual_seed(seed)
    seeds = (torch.rand((n_seeds, 3), generator=g, device=dev) - 0.5) * 2.56
    cls = torch.randint(0, n_classes, (n_seeds,), generator=g, device=dev).float()
    f, h, w = depth.shape
    vv, uu = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    out = torch.empty((f, h, w), dtype=torch.float32, device=dev)
    for i in range(f):
        k, p = ks[i], poses[i]
        x = (uu - k[0, 2]) / k[0, 0] * depth[i]
        y = (vv - k[1, 2]) / k[1, 1] * depth[i]
        pts = torch.stack((x, y, depth[i]), dim=-1).reshape(-1, 3) @ p[:3, :3].T + p[:3, 3]
        out[i] = cls[torch.cdist(pts, seeds).argmin(dim=1)].view(h, w)
    return out


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process tree
    (torch.distributed.run) before this process has touched the GPU, forward rank 0's JSON line, and
    exit with the launcher's code.  (A process that has initialised HIP must never exec or fork ranks.)"""
    import socket
    import subprocess

    if os.environ.get("SAF_BENCH_ONE_DEVICE") != "1":
        have = torch.cuda.device_count()  # counting devices does not initialise the GPU
        if have < a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but this node shows {have} GPU(s) "
                             "(SAF_BENCH_ONE_DEVICE=1 --backend gloo rehearses the control flow on one GPU)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or f"the {a.gpus}-rank run printed no result line")
    print(line, flush=True)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if world > 1 and a.feat_dtype != "f32":
        raise SystemExit("--gpus > 1 needs --feat-dtype f32: the per-rank volumes hold SUMS until the merge "
                         "(spatially_aware_ai_amd/distributed.py); bf16 sums would round every addition")
    if a.query:
        return bench_query(a, world, rank, local_rank)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    if a.side_traffic:
        side_traffic(a, a.side_traffic)
        return
    if a.scene:
        torch.cuda.set_device(local_rank)
        r = bench_scene(a, torch.device("cuda", local_rank), a.scene_frames)
        print(json.dumps({"metric": "scene latency: scan -> volume, objects, mesh, artefacts, first text query", "value": r["total"],
                          "unit": "s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": round(r["total"] * 1e3, 1),
                          "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": r["workload"]}, "scene_latency_s": r}), flush=True)
        return
    if os.environ.get("SAF_BENCH_ONE_DEVICE") == "1":
        local_rank = 0  # rehearsal: every rank on the one GPU of the box
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(a.backend)
        assert dist.get_world_size() == world

    from spatially_aware_ai_amd import ClipFusion, ClipSeemFusion
    from spatially_aware_ai_amd import distributed as sdist

    npy, npx = syn.feature_map_shape(a.width, a.height)
    grid = syn.make_grid(a.grid3, side=2.56 * a.grid3[0] / max(a.grid3))
    n_vox = grid.n_voxels

    class ResidentFeatures:  # stands in for the CLIP backbone: feature maps are already in HBM
        feature_dim = a.dim

    fdt = torch.bfloat16 if a.feat_dtype == "bf16" else torch.float32
    esz = 2 if a.feat_dtype == "bf16" else 4

    def new_volume(nvox=None, index_offset=(0, 0, 0), x_planes=None):
        nvox = grid.nvox if nvox is None else nvox
        if a.labels:
            fz = ClipSeemFusion(grid.origin, grid.voxel_size, nvox, grid.trunc, False, a.height // 3,
                                a.height // 6, ResidentFeatures(), None, keep_xyz_world=False, feat_dtype=fdt,
                                index_offset=index_offset, x_planes=x_planes)
        else:
            fz = ClipFusion(grid.origin, grid.voxel_size, nvox, grid.trunc, False, ResidentFeatures(), None,
                            a.height // 3, a.height // 6, keep_xyz_world=False, feat_dtype=fdt, index_offset=index_offset,
                            x_planes=x_planes)
        return fz.to(device)

    # N > 1 keeps two volumes: the second one lets job k+1 fuse while job k's merge drains (second timed
    # region) and holds the single-rank reference of the integrity pass
    fusions = [new_volume() for _ in range(2 if world > 1 else 1)]
    fusion = fusions[0]

    uniq = min(a.unique_frames, a.frames)
    depth, rgb, poses, ks, feat = gen_frames_gpu(uniq, a.width, a.height, a.dim, npy, npx, a.depth_kind,
                                                 1000 + rank, device, pose_kind=a.pose_kind)
    label_maps = None
    if a.labels:
        if a.label_kind == "world":
            label_maps = world_label_maps(depth, poses, ks)
        else:
            gl = torch.Generator(device=device).manual_seed(77 + rank)
            label_maps = torch.randint(0, 134, (uniq, a.height, a.width), generator=gl, device=device).float()
    arr_u, keep, _, _ = fusion._make_frames(depth, rgb, poses, ks, feat, label_maps, a.labels)
    frames = (_abi.SafFrame * a.frames)()
    for i in range(a.frames):
        frames[i] = arr_u[i % uniq]
    ws = fusion._get_workspace(npy, npx, (a.height, a.width))
    main_stream = torch.cuda.current_stream()
    stream = main_stream.cuda_stream
    comm_stream = torch.cuda.Stream() if world > 1 else main_stream
    merge_state = {"mode": a.merge, "fallback": None, "sparse": 0.0, "sparse_note": None, "last_merge": None}
    L = lib()
    prof = None
    if not a.no_profile_events:
        prof = L.saf_profiler_create(3 * a.frames * max(1, a.steps))
        L.saf_profiler_set_stride(prof, a.profile_stride)

    stats_ptr = fusion._buffers["fuse_stats"].data_ptr()

    def fuse_into(fz, frame_arr, n_frames, profiler):
        vol = fz._c_volume(for_fuse=True)  # (neither the frame queue nor the deferred clear of reset() is resolved here)
        if fz._feat_stale:
            # the volume was reset() without clearing its feature rows: the rows still unwritten after these frames are zeroed
            # inside the call (beside the last window's row kernel), as ClipFusion._fuse_now does
            rc = L.saf_fuse_frames_recycled(C.byref(vol), frame_arr, n_frames, ws.data_ptr(), ws.numel(), stats_ptr, profiler, stream)
            check(rc, "saf_fuse_frames_recycled")
            fz.__dict__["_feat_stale"] = False
            return
        rc = L.saf_fuse_frames_profiled(C.byref(vol), frame_arr, n_frames, ws.data_ptr(), ws.numel(),
                                        stats_ptr, profiler, stream)
        check(rc, "saf_fuse_frames_profiled")

    if world > 1 and merge_state["mode"] == "reduce_scatter":
        # the collectives of the merge are tried on a small tensor FIRST: a collective that raised in the middle of a
        # volume could not be followed by another one (rows already summed would be summed twice), so the real merge is
        # never retried
        why = sdist.probe_collectives(device)
        flag = torch.tensor([0 if why is None else 1], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            merge_state["fallback"] = why or "another rank's probe failed"
            merge_state["mode"] = "all_reduce"
        else:
            # rows no rank touched do not travel (distributed._merge_rows): pieces of the stripe plan whose touched share is at
            # most the threshold go packed through all_to_all_single -- probed first as well; dense throughout if it fails
            why_a2a = sdist.probe_all_to_all(device)
            merge_state["sparse"] = sdist.sparse_threshold() if why_a2a is None else 0.0  # (explicit: the library does not probe again)
            merge_state["sparse_note"] = why_a2a

    def merge(fz):
        out = sdist.merge_volumes(fz, mode=merge_state["mode"], sparse=merge_state["sparse"])
        merge_state["last_merge"] = dict(sdist.last_merge, threshold=sdist.sparse_threshold(merge_state["sparse"]))
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_region(n_steps, overlap, profiler):
        """n_steps whole jobs; returns the max-over-ranks wall time.  overlap: job k's merge runs on the side
        stream while job k+1 fuses into the other volume (events order the reuse of a volume behind its merge)."""
        merge_done = [None] * len(fusions)
        barrier()
        t0 = time.perf_counter()
        for k in range(n_steps):
            slot = k % len(fusions) if overlap else 0
            fz = fusions[slot]
            if merge_done[slot] is not None:  # this volume's previous merge must have drained
                main_stream.wait_event(merge_done[slot])
            # reset(): the small buffers are zeroed, the 34 GB of feature rows are NOT -- the windowed path never reads a
            # row whose weight is 0 -- and the rows still unwritten at the end of the job are zeroed by the fusing call itself
            # (saf_fuse_frames_recycled), inside the timed region: every buffer ends bit-identical to an up-front clear
            fz.reset(accum_mode=_abi.SAF_SUM if world > 1 else _abi.SAF_RUNNING_MEAN)
            fuse_into(fz, frames, a.frames, profiler)
            if world == 1:
                fz.flush()
            if world > 1:
                if overlap:
                    fused = main_stream.record_event()
                    with torch.cuda.stream(comm_stream):
                        comm_stream.wait_event(fused)
                        merge(fz)
                        merge_done[slot] = comm_stream.record_event()
                else:
                    merge(fz)
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_

    # ---- timed region 1 (the headline): serial merge -- fuse, then the collective, nothing overlapped ----
    run_region(a.warmup, False, None)
    fusion.fuse_stats.zero_()
    dt = run_region(a.steps, False, prof)
    st = fusion.stats()
    total_frames = world * a.frames * a.steps
    value = total_frames / dt
    # integrity of the timed work (not timed): every frame of every step was fused, and the volume of the last
    # step holds exactly one observation per valid (voxel, frame) pair -- over all ranks after the merge
    assert st["frames"] == a.frames * a.steps, f"fused {st['frames']} frames, expected {a.frames * a.steps}"
    merge_check = None
    if world == 1:
        w_sum = int(fusion.weight.sum(dtype=torch.int64))
        assert w_sum * a.steps == st["valid"], f"weight sum {w_sum} x {a.steps} steps != valid count {st['valid']}"
    else:
        own = fusion._shard_stripes if fusion._shard_stripes is not None else [(0, n_vox)]
        sums = torch.stack([sum(fusion.weight[f:f + c].sum(dtype=torch.int64) for f, c in own),
                            sum(fusion.tsdf_weight[f:f + c].sum(dtype=torch.int64) for f, c in own),
                            torch.tensor(st["valid"], device=device), torch.tensor(st["tsdf_valid"], device=device)])
        dist.all_reduce(sums)
        w_sum, tw_sum, valid_all, tsdf_all = (int(v) for v in sums.tolist())
        if fusion._shard_stripes is None:  # all_reduce: every rank holds the whole volume
            w_sum //= world
            tw_sum //= world
        assert w_sum * a.steps == valid_all, f"merged weight sum {w_sum} x {a.steps} steps != valid count of all ranks {valid_all}"
        assert tw_sum * a.steps == tsdf_all, f"merged tsdf_weight sum {tw_sum} x {a.steps} != {tsdf_all}"
        merge_check = {"weight_sum_all_ranks": w_sum, "valid_hits_all_ranks_per_step": valid_all // a.steps,
                       "tsdf_weight_sum_all_ranks": tw_sum}

    # ---- timed region 2 (N > 1): the same jobs with each merge overlapped with the next job's fusion ----
    overlapped = None
    if world > 1 and not a.no_overlap_merge:
        run_region(min(2, max(1, a.warmup)), True, None)
        dt2 = run_region(a.steps, True, None)
        overlapped = {"value": round(total_frames / dt2, 2), "unit": "frames/s", "ms_per_step": round(dt2 / a.steps * 1e3, 3),
                      "note": "job k's merge on a side stream beside job k+1's fusion into a second volume: a stream of "
                              "independent jobs, not BASELINE config 4's single job (that is `value`)"}

    # ---- timed region 3 (N > 1): the same job VOXEL-sharded -- every rank fuses every frame into its x-slab, no merge ----
    voxel_sharded = None
    if world > 1 and not a.no_voxel_sharded and int(grid.nvox[0]) >= world:
        voxel_sharded = bench_voxel_sharded(a, dist, sdist, grid, fusions, (depth, rgb, poses, ks, feat, label_maps), new_volume,
                                            world, rank, device, L, stream, npy, npx)

    # ---- timed region 4 (N > 1): ONE job with its merge pipelined slab by slab (BASELINE config 4's own layout) ----
    slab_pipe = None
    if world > 1 and merge_state["mode"] in ("reduce_scatter", "all_reduce"):
        n_slabs = 8

        def pipe_job():
            fusion.reset(accum_mode=_abi.SAF_SUM)
            return sdist.fuse_merge_pipelined(fusion, frames, a.frames, ws, n_slabs=n_slabs, comm_stream=comm_stream,
                                              mode=merge_state["mode"], stats_ptr=stats_ptr, sparse=merge_state["sparse"])

        pipe_job()
        barrier()
        fusion.fuse_stats.zero_()
        t0 = time.perf_counter()
        stripes = None
        for _ in range(a.steps):
            stripes = pipe_job()
        barrier()
        dt4 = time.perf_counter() - t0
        tmax = torch.tensor([dt4], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt4 = float(tmax.item())
        st4 = fusion.stats()
        # integrity: over all ranks, the weights of the owned stripes add up to the valid hits of all ranks' frames
        own_w = sum(int(fusion.weight[f:f + c].sum(dtype=torch.int64)) for f, c in stripes)
        sums = torch.tensor([own_w, st4["valid"]], dtype=torch.int64, device=device)
        dist.all_reduce(sums)
        w_all, v_all = (int(x) for x in sums.tolist())
        if merge_state["mode"] == "all_reduce":
            w_all //= world
        assert w_all * a.steps == v_all, f"slab-pipelined merge: owned weights {w_all} x {a.steps} steps != valid hits {v_all}"
        slab_pipe = {"value": round(total_frames / dt4, 2), "unit": "frames/s", "ms_per_step": round(dt4 / a.steps * 1e3, 3),
                     "slabs": n_slabs, "merge_check": {"owned_weight_sum_all_ranks": w_all, "valid_hits_all_ranks_per_step": v_all // a.steps},
                     "note": "the same single job as `value` (config 4's layout: frames sharded, one per-rank volume, one merge) with "
                             "the merge issued slab by slab on a communication stream while the next slab is fused; rank k ends "
                             "with the k-th part of every slab"}
        fusion.fuse_stats.zero_()

    # ---- N > 1: the headline is the better of the two forms of BASELINE config 4's job (frames sharded, one per-rank volume,
    #      one merge per job) that have both just been timed over the same K steps and proved: merged after the fusion, or
    #      merged slab by slab behind it.  Both are reported.
    headline_region, serial_merge = None, None
    if world > 1:
        headline_region = "serial_merge"
        serial_merge = {"value": round(value, 2), "unit": "frames/s", "ms_per_step": round(dt / a.steps * 1e3, 3),
                        "note": "fuse, then the collective, nothing overlapped"}
        if slab_pipe is not None and slab_pipe["value"] > value:
            headline_region = "slab_pipelined_merge"
            value = slab_pipe["value"]
            dt = dt4

    # ---- N > 1: the merged shard of a small sharded job against a single-rank fusion of ALL its frames ----
    if world > 1 and a.check_frames > 0:
        pipe_fn = None
        if slab_pipe is not None:
            pipe_fn = lambda fz, fr, c: sdist.fuse_merge_pipelined(fz, fr, c, ws, n_slabs=slab_pipe["slabs"], comm_stream=comm_stream,
                                                                  mode=merge_state["mode"], stats_ptr=stats_ptr,
                                                                  sparse=merge_state["sparse"])
        merge_check.update(check_merge(a, dist, sdist, fusions, fuse_into, merge, frames, (depth, rgb, poses, ks, feat, label_maps),
                                       min(a.check_frames, uniq), world, rank, device, pipelined=pipe_fn))
        merge_check["mode"] = merge_state["mode"]

    # ---- roofline of the dominant kernel (fuse_kernel), this rank ----
    roofline = None
    breakdown = None
    if prof:
        ms = {}
        for cls, name in ((0, "prep"), (1, "sweep"), (2, "fuse")):
            tot, n = C.c_double(0), C.c_int64(0)
            check(L.saf_profiler_read(prof, cls, C.byref(tot), C.byref(n)), "saf_profiler_read")
            ms[name] = (tot.value, n.value)
        L.saf_profiler_destroy(prof)
        n_launch = max(1, ms["fuse"][1])
        nv_per = st["valid"] / max(1, st["frames"])
        nt_per = st["tsdf_valid"] / max(1, st["frames"])
        lab = 8 if a.labels else 0  # one label counter RMW per valid voxel
        windowed = st.get("window_rows", 0) > 0
        if windowed:
            # SURVEY.md §8d, launches covering a frame set S (a window of up to WIN frames):
            # B_fuse(S) = U_v*(2*D*s + 2*12 + 2*4 [+2*4]) + U_t*(2*4 + 2*4) + sum_f (H*W*(4+12[+4]) + D*npy*npx*4)
            # with U_v = rows the window read-modify-wrote, U_t = voxels whose TSDF it updated (kernel counters).
            # fuse_window_kernel's share: the rows, rgb / weight / label side, rgb + label images and the maps;
            # classify_window_kernel's share: the TSDF term and the depth images.
            n_windows = (a.frames + WIN - 1) // WIN * a.steps
            uv = st["window_rows"] / n_windows
            ut = st["window_tsdf_voxels"] / n_windows
            fpl = a.frames * a.steps / n_windows  # frames per launch
            fuse_bytes = (uv * (2 * a.dim * esz + 2 * 12 + 2 * 4 + lab)
                          + fpl * (a.height * a.width * (12 + (4 if a.labels else 0)) + a.dim * npy * npx * 4))
            sweep_bytes = ut * 16 + fpl * a.height * a.width * 4
            frame_bytes = (fuse_bytes + sweep_bytes) / fpl
        else:
            # B_fuse = Nv*(2*D*4 + 2*12 + 2*4) + Nt*16 + H*W*16 + D*npy*npx*4 per frame.
            # The fuse kernel's share: rows + rgb + weight RMW + its read of the re-laid feature map and
            # of the compact list; the Nt*16 TSDF term and the depth image belong to the sweep kernel.
            fuse_bytes = nv_per * (2 * a.dim * esz + 2 * 12 + 2 * 4 + 4 + lab) + a.dim * npy * npx * 4
            sweep_bytes = nt_per * 16 + a.height * a.width * 4 + nv_per * 4
            frame_bytes = (nv_per * (2 * a.dim * esz + 32 + lab) + nt_per * 16 + a.height * a.width * (16 + (4 if a.labels else 0))
                           + a.dim * npy * npx * 4)
        # windowed: fuse_bytes are those of ONE WINDOW, and a window's row kernel may run as several launches (the first window
        # of a call is fused slab by slab so that its classification hides behind its own rows): the time that goes with
        # those bytes is the SUM of the window's launches = total kernel time / windows
        avg_fuse_s = ms["fuse"][0] / (n_windows if windowed else n_launch) * 1e-3
        achieved = fuse_bytes / avg_fuse_s / 1e9
        # HBM traffic of this kernel: PMC counters cannot be read from inside this process, so the figure comes
        # from the committed rocprofv3 --pmc passes of this same command (tools/profile.sh, newest round first)
        traffic, traffic_source = None, None
        kname = "fuse_window_kernel" if windowed else "fuse_kernel"
        # (the order-free D = 512 instantiations -- the benchmark's -- are wrappers of their own around the common body since
        #  round 6: that is the name in rocprofv3's tables, profiles/r06/kernel_stats_headline.csv)
        kname_prof = "fuse_window_kernel_of176" if windowed and a.dim == 512 and os.environ.get("SAF_WIN_FORM", "s")[0] == "s" else kname
        for rnd in ("r04", "r03", "r02", "r01"):
            rel = os.path.join("profiles", rnd, "window_traffic.json" if windowed else "fuse_traffic.json")
            try:
                tj = json.load(open(os.path.join(ROOT, rel)))
            except Exception:
                continue
            if (tj.get("grid") == a.grid and tj.get("dim") == a.dim and tj.get("depth_kind", "A") == a.depth_kind
                    and tj.get("dtype", "f32") == a.feat_dtype and not a.labels
                    and tj.get("frames_per_launch", 1) == (WIN if windowed else 1)):
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = (f"{rel}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on an earlier box "
                                  "(FETCH_SIZE x2, gfx950), not measured in this run")
                break
        roofline = {
            "kernel": kname, "kernel_name_in_profiles": kname_prof, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "volume_state": "fresh (every step starts from a zeroed volume: a row is not read in the window that first "
                            "touches it, but its bytes are counted as algorithmic)",
            "algorithmic_bytes_per_launch": int(fuse_bytes), "avg_launch_us": round(avg_fuse_s * 1e6, 2),
            "launches": int(n_launch), "windows": int(n_windows) if windowed else None,
            "valid_voxels_per_frame": round(nv_per, 1),
            "tsdf_voxels_per_frame": round(nt_per, 1),
        }
        if windowed:
            roofline["frames_per_launch"] = round(fpl, 1)
            roofline["rows_per_launch"] = round(uv, 1)
            roofline["tsdf_voxels_per_launch"] = round(ut, 1)
            roofline["hits_per_row"] = round(st["valid"] / max(1, st["window_rows"]), 3)
            # SURVEY.md 8d's bytes of the SAME frames fused one launch per frame (U_v = Nv, U_t = Nt): what a design that
            # moves every frame's rows to HBM and back could at best reach at the HBM peak
            single = (nv_per * (2 * a.dim * esz + 2 * 12 + 2 * 4 + lab) + nt_per * 16
                      + a.height * a.width * (16 + (4 if a.labels else 0)) + a.dim * npy * npx * 4)
            roofline["frame_at_a_time"] = {
                "algorithmic_bytes_per_frame": int(single),
                "frames_per_s_at_hbm_peak": round(HBM_PEAK_GBS * 1e9 / single, 1),
                "windowed_bytes_per_frame": int(frame_bytes),
                "this_run_frames_per_s_per_gpu": round(a.frames * a.steps / dt, 1),
                "frac_of_that_bound": round(a.frames * a.steps / dt * single / 1e9 / HBM_PEAK_GBS, 4),
                "note": "SURVEY.md 8d prices the fuse at these bytes (its '40 %% target' = 0.4 x frames_per_s_at_hbm_peak); "
                        "the window form moves %.0f %% of the bytes a frame-at-a-time fusion must move (rows travel once per "
                        "window of %d frames instead of once per frame): `frac` prices the bytes this kernel is left with, "
                        "so it FALLS when a longer window removes bytes faster than time" % (100.0 * frame_bytes / single, WIN)}
            try:  # what the kernel is bound by: the L2 -> L1 path with HBM-latency requests mixed in (committed PMC pass + probe)
                pj = json.load(open(os.path.join(ROOT, "profiles", "r04", "rows_pmc.json")))["rows"]
                if a.grid == 256 and a.dim == 512 and a.depth_kind == "A" and a.feat_dtype == "f32" and not a.labels and WIN == 128:
                    gb = pj["TCP_TCC_READ_REQ_sum"] * 128.0
                    roofline["l2_gather"] = {
                        "bytes_per_launch": int(gb), "launch_us": round(pj["us_pass_e"], 1),
                        "achieved_TBps": round(gb / pj["us_pass_e"] / 1e6, 2),
                        "ceiling_TBps": {"L2-resident 2 KiB rows alone (tools/gather_probe.hip)": [32.2, 34.4],
                                         "the same with one HBM row read and written back per two batches of 16 KiB (5.9 % of the "
                                         "requests; this kernel has one per 1.4 batches)": [16.1, 18.6]},
                        "source": "profiles/r04/rows_pmc.json (rocprofv3 --pmc TCP_TCC_READ_REQ_sum x 128 B over the kernel's duration in "
                                  "that pass: the four windows of one 512-frame job, nothing beside the kernel, earlier box); ceilings: "
                                  "profiles/r04/gather_probe.log, gather_probe_mix.log -- 8, 12 or 16 waves per CU alike"}
            except Exception:
                pass
            form = os.environ.get("SAF_WIN_FORM", "sums")
            roofline["form"] = form
            roofline["note"] = ("voxel-major window kernel: reads and writes every touched feature row once per window of %d " % WIN +
                                ("frames (order-free form: a row's samples of the window are summed in registers and blended once -- "
                                 "feature values within fp32 rounding of frame-by-frame fusion, everything else bit-identical)"
                                 if form[0] == "s" else "frames (hits applied in frame order: bit-identical to frame-by-frame fusion)") +
                                "; `avg_launch_us` = the row kernel's time per WINDOW (one launch per window unless SAF_WIN_SLABS / "
                                "SAF_WIN_W0_SLABS cut windows into slabs: `launches` / `windows`); the NEXT "
                                "unit's classification + TSDF (classify kernels, kernel_breakdown.sweep_us per launch of 32 frames) "
                                "run beside it on a second stream, so this duration is that of a kernel sharing the chip: `isolated` "
                                "is the same kernel alone")
        breakdown = {
            "prep_us": round(ms["prep"][0] / max(1, ms["prep"][1]) * 1e3, 2),
            "sweep_us": round(ms["sweep"][0] / max(1, ms["sweep"][1]) * 1e3, 2),
            "fuse_us": round(avg_fuse_s * 1e6, 2),
            "sweep_algorithmic_bytes": int(sweep_bytes),
            "frame_algorithmic_bytes": int(frame_bytes),
            "frame_hbm_frac": round(frame_bytes / (dt / (a.frames * a.steps)) / 1e9 / HBM_PEAK_GBS, 4),
        }

    # ---- the row kernel with nothing beside it (the timed region runs the NEXT window's classification next to it) ----
    if roofline is not None and windowed and rank == 0 and world == 1:
        prof4 = L.saf_profiler_create(3 * a.frames)
        L.saf_profiler_set_stride(prof4, a.profile_stride)
        os.environ["SAF_WIN_OVERLAP"] = "0"  # read per call by saf_fuse_frames: every kernel on the caller's stream
        try:
            fusion.reset()
            fuse_into(fusion, frames, a.frames, prof4)
            torch.cuda.synchronize()
        finally:
            del os.environ["SAF_WIN_OVERLAP"]
        tot, n = C.c_double(0), C.c_int64(0)
        check(L.saf_profiler_read(prof4, 2, C.byref(tot), C.byref(n)), "saf_profiler_read")
        tot_c, n_c = C.c_double(0), C.c_int64(0)
        check(L.saf_profiler_read(prof4, 1, C.byref(tot_c), C.byref(n_c)), "saf_profiler_read")
        L.saf_profiler_destroy(prof4)
        if n.value:
            iso = tot.value / n.value * 1e-3
            roofline["isolated"] = {"avg_launch_us": round(iso * 1e6, 2), "achieved": round(fuse_bytes / iso / 1e9, 1),
                                    "frac": round(fuse_bytes / iso / 1e9 / HBM_PEAK_GBS, 4), "launches": int(n.value),
                                    "classify_us_alone": round(tot_c.value / max(1, n_c.value) * 1e3, 2),
                                    "note": "SAF_WIN_OVERLAP=0: classification and row kernel one after the other on one stream; "
                                            "in the timed region window w + 1 is classified beside window w's row kernel, "
                                            "which lengthens each kernel and shortens the step"}
        fusion.fuse_stats.zero_()

    # ---- the same launches into a WARM volume (no zeroing: every touched row is read), untimed in `value` ----
    if roofline is not None and windowed and rank == 0 and world == 1:
        prof3 = L.saf_profiler_create(3 * a.frames)
        L.saf_profiler_set_stride(prof3, a.profile_stride)
        fusion.fuse_stats.zero_()
        fuse_into(fusion, frames, a.frames, prof3)  # the volume still holds the last step's 512 frames
        torch.cuda.synchronize()
        st3 = fusion.stats()
        tot, n = C.c_double(0), C.c_int64(0)
        check(L.saf_profiler_read(prof3, 2, C.byref(tot), C.byref(n)), "saf_profiler_read")
        L.saf_profiler_destroy(prof3)
        if n.value:
            nwin = (a.frames + WIN - 1) // WIN
            wb = (st3["window_rows"] / nwin * (2 * a.dim * esz + 2 * 12 + 2 * 4 + lab)
                  + a.frames / nwin * (a.height * a.width * (12 + (4 if a.labels else 0)) + a.dim * npy * npx * 4))
            ws_ = tot.value / nwin * 1e-3  # (per window: the first window runs as several launches)
            roofline["warm_volume"] = {"avg_launch_us": round(ws_ * 1e6, 2), "achieved": round(wb / ws_ / 1e9, 1),
                                       "frac": round(wb / ws_ / 1e9 / HBM_PEAK_GBS, 4), "launches": int(n.value),
                                       "algorithmic_bytes_per_launch": int(wb),
                                       "note": "the same frames fused again without zeroing: every touched row is read and written"}

    # ---- the same kernel timed alone (one frame per call = no sweep running beside it) ----
    if roofline is not None and rank == 0 and not windowed:
        n_iso = min(32, a.frames)
        prof2 = L.saf_profiler_create(3 * n_iso)
        torch.cuda.synchronize()
        for i in range(n_iso):
            vol_i = fusion._c_volume(for_fuse=True)
            check(L.saf_fuse_frames_profiled(C.byref(vol_i), C.byref(frames[i]), 1, ws.data_ptr(), ws.numel(),
                                             fusion.fuse_stats.data_ptr(), prof2, stream), "isolated pass")
        torch.cuda.synchronize()
        tot, n = C.c_double(0), C.c_int64(0)
        check(L.saf_profiler_read(prof2, 2, C.byref(tot), C.byref(n)), "saf_profiler_read")
        tot_s, n_s = C.c_double(0), C.c_int64(0)
        check(L.saf_profiler_read(prof2, 1, C.byref(tot_s), C.byref(n_s)), "saf_profiler_read")
        L.saf_profiler_destroy(prof2)
        iso_s = tot.value / max(1, n.value) * 1e-3
        roofline["isolated"] = {
            "avg_launch_us": round(iso_s * 1e6, 2), "achieved": round(fuse_bytes / iso_s / 1e9, 1),
            "frac": round(fuse_bytes / iso_s / 1e9 / HBM_PEAK_GBS, 4), "launches": int(n.value),
            "sweep_us": round(tot_s.value / max(1, n_s.value) * 1e3, 2),
            "note": "fuse kernel alone on the chip (one frame per call); in the timed region it shares the chip "
                    "with the sweeps of the following frames",
        }

    # ---- the reference's call pattern: one frame per integrate() call (clipfusion.py:1125-1133) ----
    # (right behind the timed region and the isolated pass, on the same volume: `vs_bulk` compares like with like -- behind the
    #  end-to-end pass's ViT the same loop measures 4 % lower, the chip's clock not yet back)
    api_b1 = None
    if a.api_b1 > 0 and rank == 0 and world == 1:
        api_b1 = bench_api_b1(a, fusion, depth, rgb, poses, ks, feat, label_maps, value)
        # the same loop with `fusion.borrow_inputs = True`: the queue reads the calls' depth / rgb images where they lie instead of
        # copying them into its ring (the caller does not write them before flush(): the reference's loop never does)
        fusion.borrow_inputs = True
        try:
            bb = bench_api_b1(a, fusion, depth, rgb, poses, ks, feat, label_maps, value)
            api_b1["borrowed_inputs"] = {k: bb[k] for k in ("value", "vs_bulk", "host_enqueue_us_per_call")}
        finally:
            fusion.borrow_inputs = False

    # ---- end-to-end: backbone + fuse through the reference-shaped Python API (reported separately) ----
    e2e = None
    if a.end_to_end > 0 and rank == 0 and world == 1 and a.dim == 512 and not a.labels:
        from spatially_aware_ai_amd.backbones import RandomViTB32
        from spatially_aware_ai_amd.clipfusion import Clip

        clip = Clip("ViT-B-32 (random weights)", None, backbone=RandomViTB32(), tokenizer=None).to(device).eval()
        clip.requires_grad_(False)
        if a.e2e_tile_batch > 0:
            clip.max_patch_batch_size = a.e2e_tile_batch
        fz = ClipFusion(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, False, clip, None, a.height // 3,
                        a.height // 6, keep_xyz_world=False, feat_dtype=fdt).to(device)
        nfr = min(a.end_to_end, uniq)
        bs = max(1, a.e2e_batch)

        def run_e2e():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.e2e_dtype == "bf16"):
                for s0 in range(0, nfr, bs):
                    sl = slice(s0, min(nfr, s0 + bs))
                    fz.integrate(depth[sl], rgb[sl], poses[sl], ks[sl])
                fz.flush()  # small calls are queued (frames and backbone): the job ends when the volume is up to date

        run_e2e()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_e2e()
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.e2e_dtype == "bf16"):
            t2 = time.perf_counter()
            for s0 in range(0, nfr, bs):
                clip.img_inference_tiled(rgb[s0 : s0 + bs].permute(0, 3, 1, 2), a.height // 3, a.height // 6)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t2
        e2e = {
            "value": round(nfr / d1, 1), "unit": "frames/s", "frames": nfr, "batch": bs,
            "tiles_per_encode_call": int(clip.max_patch_batch_size),
            "backbone": "ViT-B/32 image tower, seeded random weights (no CLIP weights offline), 35 tiles/frame, "
                        + a.e2e_dtype,
            "backbone_only_frames_per_s": round(nfr / d2, 1),
            "backbone_deferred_to_flush": bool(bs <= fz._DEFER_MAX_BATCH),
            "note": "Clip.img_inference_tiled (PyTorch-ROCm) + saf_fuse_frames through ClipFusion.integrate; calls of up to 15 "
                    "frames are queued and the ViT runs on all queued frames when the queue is flushed (128 frames x 35 tiles, "
                    "1024 tiles per encode_image call); kMaX is not part of this pass",
        }

    # ---- N = 1: the job fused slab by slab (what the slab-pipelined merge of N > 1 does between its collectives) ----
    slabwise = None
    if rank == 0 and world == 1 and windowed_headline(st) and not a.no_side:
        n_slabs = 8
        bounds = sdist.slab_bounds(int(grid.nvox[0]), n_slabs, ramp=True)
        x0s = (C.c_int32 * len(bounds))(*[b[0] for b in bounds])
        nxs = (C.c_int32 * len(bounds))(*[b[1] for b in bounds])

        def whole():
            fusion.reset()
            fusion.flush()
            fuse_into(fusion, frames, a.frames, None)

        def by_slab():
            fusion.reset()
            fusion.flush()
            vol = fusion._c_volume(for_fuse=True)
            check(L.saf_fuse_frames_slabs(C.byref(vol), frames, a.frames, x0s, nxs, len(bounds), None, 0, ws.data_ptr(), ws.numel(),
                                          stats_ptr, None, stream), "slab-wise fuse")

        def timed2(fn):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 2

        t_whole, t_slab = timed2(whole), timed2(by_slab)
        fusion.fuse_stats.zero_()
        slabwise = {"slabs": len(bounds), "ms_whole_volume": round(t_whole * 1e3, 2), "ms_slab_by_slab": round(t_slab * 1e3, 2),
                    "ratio": round(t_slab / t_whole, 3),
                    "planes_per_slab": [b[1] for b in bounds],
                    "note": "the rank's 512 frames fused into x-slabs of the ONE per-rank volume by saf_fuse_frames_slabs (one call: a "
                            "slab's first window is classified beside the last row kernel of the slab before it; small slabs at both "
                            "ends; up-front clear in both): behind every slab's event the N > 1 job starts that slab's reduce-scatter"}

    # ---- the other configurations and the copy rate of this box, measured in this run (rank 0, N = 1) ----
    side, copy_rate = None, None
    if rank == 0 and world == 1 and not a.no_side and isinstance(a.grid, int) and a.grid == 256 and a.depth_kind == "A" and a.pose_kind == "look_at" and not a.labels \
            and a.feat_dtype == "f32" and a.width == 640 and a.height == 480:
        copy_rate = hbm_copy_rate(device)
        side = side_workloads(a, device, L, (depth, rgb, poses, ks, feat), npy, npx)
    if roofline is not None and copy_rate is not None:
        roofline["hbm_copy_GBps"] = copy_rate
        roofline["frac_of_copy_rate"] = round(roofline["achieved"] / copy_rate, 4)
    # ---- roofline.traffic measured in THIS run (rank 0, N = 1, the default command): PMC counters cannot be read in-process, so
    #      one job is run twice more as a child under rocprofv3 --pmc (FETCH_SIZE, then WRITE_SIZE: separate passes)
    if roofline is not None and side is not None and not a.no_pmc and roofline.get("kernel") == "fuse_window_kernel":
        m = measure_traffic(a)
        if m is not None:
            roofline["traffic"] = m["hbm_bytes_per_launch"]
            roofline["traffic_source"] = m["source"]
            roofline["traffic_read_bytes"], roofline["traffic_write_bytes"] = m["read"], m["write"]

    # ---- CPU baseline: the oracle on a bounded sample of the same frames (rank 0, N=1 only) ----
    cpu = None
    if rank == 0 and world == 1 and a.cpu_frames != 0:
        cpu = cpu_baseline(a, grid, depth, rgb, poses, ks, feat, npy, npx)

    if rank == 0:
        out = {
            "metric": f"fused RGB-D frames/sec into {a.grid}^3x{a.dim} voxel grid",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.feat_dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{a.frames} frames/rank {a.width}x{a.height} depth-{a.depth_kind}{' free (rolled, fx != fy) poses' if a.pose_kind == 'free' else ''}, per-rank "
                            f"{a.grid}^3x{a.dim} {a.feat_dtype} grid{' + panoptic label histogram' if a.labels else ''}, frames sharded, "
                            + ((f"one {a.backend} {merge_state['mode']} merge per step, "
                                + ("issued slab by slab behind the fusion" if headline_region == "slab_pipelined_merge"
                                   else "serial (fuse, then merge)"))
                               if world > 1 else "single GPU (no merge)"),
                "frames_per_rank": a.frames, "grid": a.grid, "feat_dim": a.dim, "image": [a.width, a.height],
                "feature_map": [npy, npx], "unique_frames_resident": uniq, "n_voxels": n_vox,
                "parallelism": f"frames-dp{world}", "merge_fallback": merge_state["fallback"],
                "merge_sparse_route": ({"pieces_packed_of": [merge_state["last_merge"]["packed"], merge_state["last_merge"]["pieces"]],
                                        "touched_rows": merge_state["last_merge"]["touched_rows"], "rows": merge_state["last_merge"]["rows"],
                                        "threshold": merge_state["last_merge"]["threshold"], "probe": merge_state["sparse_note"]}
                                       if merge_state["last_merge"] else None),
                "rccl_world": dist.get_world_size() if world > 1 else 1, "backend": a.backend if world > 1 else None,
            },
            "headline_region": headline_region,
            "serial_merge": serial_merge,
            "overlapped_merge": overlapped,
            "voxel_sharded": voxel_sharded,
            "merge_check": merge_check,
            "roofline": roofline,
            "kernel_breakdown": breakdown,
            "cpu_baseline": cpu,
            "end_to_end": e2e,
            "hbm_copy_GBps": copy_rate,
            "slab_by_slab_fuse": slabwise,
            "slab_pipelined_merge": slab_pipe,
            "side_workloads": side,
        }
        if api_b1 is not None:
            out["api_b1"] = api_b1
            if side is not None:  # (the driver's default line: the reference's own call pattern beside the other workloads)
                side["api_b1"] = api_b1
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()



def windowed_headline(st):
    return st.get("window_rows", 0) > 0


def measure_traffic(a, timeout_s=150, overrides=None):
    """HBM bytes per window of fuse_window_kernel, measured now: this script runs ONE job (512 frames, no warm-up, no side work)
    as a child process under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes; output under
    /tmp), the per-launch counter values of the row kernel are averaged.  FETCH_SIZE is doubled (on gfx950 it reports half the
    bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM section), both are KiB.  None if the profiler is not there or a
    pass fails -- the committed figure stays in the line then."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None  # this process is itself being profiled (tools/profile.sh): no profiler inside a profiler
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="saf_pmc_", dir="/tmp")
        o = dict(grid=str(a.grid), depth_kind=a.depth_kind, feat_dtype=a.feat_dtype, labels=a.labels, label_kind=a.label_kind,
                 pose_kind=a.pose_kind)
        o.update(overrides or {})
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
               os.path.abspath(__file__), "--cpu-frames", "0", "--steps", "1", "--warmup", "0", "--no-profile-events", "--no-side",
               "--end-to-end", "0", "--no-pmc", "--frames", str(a.frames), "--grid", o["grid"], "--dim", str(a.dim),
               "--depth-kind", o["depth_kind"], "--feat-dtype", o["feat_dtype"], "--label-kind", o["label_kind"],
               "--pose-kind", o["pose_kind"], "--api-b1", "0"] + (
                   ["--labels"] if o["labels"] else [])
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout_s, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            per = []
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "fuse_window_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        per.append(float(r["Counter_Value"]))
            if not per:
                return None
            # per WINDOW (a window cut into slab launches -- SAF_WIN_SLABS -- is still one unit of `avg_launch_us`)
            vals[counter] = sum(per) / max(1, (a.frames + WIN - 1) // WIN)
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    read, write = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
    return {"read": int(read), "write": int(write), "hbm_bytes_per_launch": int(read + write),
            "source": "measured in this run: one job as a child process under rocprofv3 --pmc FETCH_SIZE and again under --pmc "
                      "WRITE_SIZE (separate passes), the job's row-kernel launches summed per window; FETCH_SIZE x 2 (gfx950), KiB -> bytes"}


SIDE_TRAFFIC_CASES = {  # side workload -> the main-path flags of the same job (bench.py --side-traffic)
    "config2_128cube_f32": dict(grid="128"),
    "config3_256cube_bf16_labels": dict(feat_dtype="bf16", labels=True, label_kind="iid"),
    "config3_256cube_bf16_labels_consistent": dict(feat_dtype="bf16", labels=True, label_kind="world"),
    "coherent_scene_depth_B": dict(depth_kind="B"),
    "coherent_scene_free_poses": dict(depth_kind="B", pose_kind="free"),
}


def side_traffic(a, path):
    """`bench.py --side-traffic`: the counter traffic (HBM bytes per window of the row kernel) of every side workload's job, two
    profiler passes each, written to `path` (committed as profiles/r06/side_traffic.json: the default run, which must stay
    within minutes, quotes it instead of running eight more child processes)."""
    out = {}
    for name, ov in SIDE_TRAFFIC_CASES.items():
        m = measure_traffic(a, overrides=ov)
        out[name] = None if m is None else {"read": m["read"], "write": m["write"], "hbm_bytes_per_window": m["hbm_bytes_per_launch"]}
        print(name, out[name], flush=True)
    out["method"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over one 512-frame job of the same seeds and shapes as the "
                     "side workload, row-kernel launches summed per window; FETCH_SIZE x 2 (gfx950), KiB -> bytes")
    json.dump(out, open(path, "w"), indent=1)
    return out


def hbm_copy_rate(device, nbytes=4 << 30, reps=3):
    """Device-to-device copy rate measured in this run (read + write bytes per second), beside the 8 TB/s spec peak."""
    x = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    y = torch.empty_like(x)
    y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    s = e0.elapsed_time(e1) * 1e-3 / reps
    return round(2 * nbytes / s / 1e9, 1)


def side_workloads(a, device, L, frames_A, npy, npx):
    """The other BASELINE configurations on the same box, each a short timed job with its own roofline (rank 0, N = 1):
    config 2 (128^3 x 512 f32), config 3 (256^3 bf16 + panoptic label histogram), the coherent scene (depth B), config 5
    (1000 fp16 queries over 256^3 x 512: per-voxel best query, heat maps, per-query best voxel).  About ten seconds."""
    from spatially_aware_ai_amd import ClipFusion, ClipSeemFusion
    from spatially_aware_ai_amd.clipfusion import query_scan_wide

    t_begin = time.perf_counter()
    depth, rgb, poses, ks, feat = frames_A
    uniq = depth.shape[0]
    n_frames = 512
    out = {}

    class Resident:
        feature_dim = a.dim

    def fuse_case(name, nvox, fdt, labels, fr, note, label_kind="iid"):
        grid = syn.make_grid(nvox)
        d_, r_, p_, k_, f_ = fr
        if labels:
            fz = ClipSeemFusion(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, False, a.height // 3, a.height // 6,
                                Resident(), None, keep_xyz_world=False, feat_dtype=fdt).to(device)
            if label_kind == "world":
                lm = world_label_maps(d_, p_, k_)
            else:
                gl = torch.Generator(device=device).manual_seed(77)
                lm = torch.randint(0, 134, (d_.shape[0], a.height, a.width), generator=gl, device=device).float()
        else:
            fz = ClipFusion(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, False, Resident(), None, a.height // 3,
                            a.height // 6, keep_xyz_world=False, feat_dtype=fdt).to(device)
            lm = None
        arr_u, keep, _, _ = fz._make_frames(d_, r_, p_, k_, f_, lm, labels)
        arr = (_abi.SafFrame * n_frames)()
        for i in range(n_frames):
            arr[i] = arr_u[i % d_.shape[0]]
        ws = fz._get_workspace(npy, npx, (a.height, a.width))
        stream = torch.cuda.current_stream().cuda_stream
        prof = L.saf_profiler_create(3 * n_frames)
        L.saf_profiler_set_stride(prof, 4)
        esz = 2 if fdt == torch.bfloat16 else 4

        def job(p):
            fz.reset()
            vol = fz._c_volume(for_fuse=True)
            # (reset() left the feature rows alone: the call zeroes the ones still unwritten, as ClipFusion._fuse_now does)
            check(L.saf_fuse_frames_recycled(C.byref(vol), arr, n_frames, ws.data_ptr(), ws.numel(),
                                             fz._buffers["fuse_stats"].data_ptr(), p, stream), "side workload")
            fz.__dict__["_feat_stale"] = False
            fz.flush()

        job(None)
        torch.cuda.synchronize()
        fz.fuse_stats.zero_()
        t0 = time.perf_counter()
        job(prof)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = fz.stats()
        tot, n = C.c_double(0), C.c_int64(0)
        check(L.saf_profiler_read(prof, 2, C.byref(tot), C.byref(n)), "saf_profiler_read")
        L.saf_profiler_destroy(prof)
        lab = 8 if labels else 0
        n_win = (n_frames + WIN - 1) // WIN
        uv = st["window_rows"] / n_win
        fuse_bytes = (uv * (2 * a.dim * esz + 2 * 12 + 2 * 4 + lab)
                      + n_frames / n_win * (a.height * a.width * (12 + (4 if labels else 0)) + a.dim * npy * npx * 4))
        kern = tot.value / max(1, n_win) * 1e-3  # per WINDOW (its first window runs as several slab launches)
        out[name] = {
            "value": round(n_frames / dt, 1), "unit": "frames/s", "ms_per_job": round(dt * 1e3, 2), "frames": n_frames,
            "workload": note, "valid_voxels_per_frame": round(st["valid"] / n_frames, 1),
            "hits_per_row": round(st["valid"] / max(1, st["window_rows"]), 2),
            "roofline": {"kernel": "fuse_window_kernel" if os.environ.get("SAF_WIN_FORM", "s")[0] != "b" else "fuse_brick_kernel",
                         "bound": "hbm", "achieved": round(fuse_bytes / kern / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(fuse_bytes / kern / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "avg_launch_us": round(kern * 1e6, 1), "launches": int(n.value), "windows": int(n_win),
                         "algorithmic_bytes_per_launch": int(fuse_bytes)}}
        try:  # counter traffic of the same job: measured this round with `bench.py --side-traffic` (two profiler passes per workload)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r06", "side_traffic.json")))
            if tj.get(name) and a.dim == 512 and a.width == 640 and a.height == 480:
                out[name]["roofline"]["traffic"] = tj[name]["hbm_bytes_per_window"]
                out[name]["roofline"]["traffic_source"] = "profiles/r06/side_traffic.json (this round, another box): " + tj["method"]
        except Exception:  # noqa: BLE001
            pass
        del fz, ws, keep
        torch.cuda.empty_cache()

    fuse_case("config2_128cube_f32", 128, torch.float32, False, frames_A,
              "BASELINE config 2's shape: 512 frames 640x480 (depth A) into a 128^3 x 512 fp32 grid (resident ViT-B/32-shaped features)")
    if a.dim % 512 == 0:
        fuse_case("config3_256cube_bf16_labels", 256, torch.bfloat16, True, frames_A,
                  "BASELINE config 3's fused part: 512 frames into a 256^3 x 512 bf16 grid + the 143-class label histogram "
                  "(bf16 features: the window's map images are kept in bf16, DESIGN 4.6c); panoptic maps iid per pixel (SURVEY 8d: "
                  "no two frames agree on a voxel's class -- the worst case for the histogram)")
        fuse_case("config3_256cube_bf16_labels_consistent", 256, torch.bfloat16, True, frames_A,
                  "config 3's fused part with panoptic maps that are consistent in 3-D (the class of the Voronoi cell, of 40 seeded "
                  "points, that the observed surface point lies in): what a segmentation of a static scene looks like -- a voxel's "
                  "equal classes of consecutive frames are counted with one add", label_kind="world")
    nb = min(128, uniq)
    frames_B = gen_frames_gpu(nb, a.width, a.height, a.dim, npy, npx, "B", 2000, device)
    fuse_case("coherent_scene_depth_B", 256, torch.float32, False, frames_B,
              "the headline job on the coherent analytic scene (sphere in a box, SURVEY 8d depth B): 512 frames (128 unique), 256^3 x 512 fp32")
    del frames_B
    frames_F = gen_frames_gpu(nb, a.width, a.height, a.dim, npy, npx, "B", 2001, device, pose_kind="free")
    fuse_case("coherent_scene_free_poses", 256, torch.float32, False, frames_F,
              "the coherent scene seen by ROLLED cameras with fx != fy and an off-centre principal point (synthetic.family_pose: "
              "the poses of a hand-held scan, clipfusion.py:308-312): 512 frames (128 unique), 256^3 x 512 fp32")
    del frames_F

    # ---- config 5: 1000 fp16 queries over the 256^3 x 512 volume
    n, d, q, n_bg = 256 ** 3, a.dim, 1000, 4
    g = torch.Generator(device=device).manual_seed(100)
    feats16 = torch.empty((n, d), dtype=torch.float16, device=device)
    for s0 in range(0, n, 1 << 20):
        feats16[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), d), generator=g, device=device).half()
    text = torch.randn((n_bg + q, d), generator=torch.Generator().manual_seed(9))
    text = (text / text.norm(dim=-1, keepdim=True)).to(device)
    # ---- the reference's own online scans over the fp32 volume (clip_text_query: L = 5 softmax to the last column,
    #      query_mesh.py:36-39; L = 63 surgery, :52-83): the split scan (round 6) -- fp32 scores from fp16 matrix instructions
    if d % 16 == 0:
        from spatially_aware_ai_amd.clipfusion import _query_scan
        f32 = torch.empty((n, d), dtype=torch.float32, device=device)
        for s0 in range(0, n, 1 << 20):
            f32[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), d), generator=g, device=device)
        split = os.environ.get("SAF_Q_SPLIT", "1") != "0"
        b16 = torch.empty((n, d), dtype=torch.bfloat16, device=device)  # config 3's volume dtype: the 16-bit form of the scan
        for s0 in range(0, n, 1 << 20):
            b16[s0:s0 + (1 << 20)] = f32[s0:s0 + (1 << 20)].to(torch.bfloat16)
        for name, nl, epi, scale, vol_t in (("a9_softmax_L5_fp32_volume", 5, _abi.SAF_Q_SOFTMAX, 100.0, f32),
                                            ("a10_surgery_L63_fp32_volume", 63, _abi.SAF_Q_SURGERY, 1.0, f32),
                                            ("a10_surgery_L63_bf16_volume", 63, _abi.SAF_Q_SURGERY, 1.0, b16)):
            t = text[:nl]
            esz = vol_t.element_size()
            last = epi == _abi.SAF_Q_SOFTMAX
            hold = {}
            fn = lambda: hold.__setitem__("o", _query_scan(vol_t, t, epi, scale=scale, normalize=True, last_only=last))  # noqa: E731
            fn()
            fn()  # (two calls: the second still holds the first's [N, L] output, so both of the allocator's blocks exist before the timing)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            fn()
            fn()
            e1.record()
            torch.cuda.synchronize()
            kern = e0.elapsed_time(e1) * 1e-3 / 3
            nbytes = n * d * esz + nl * d * 4 + n * (1 if last else nl) * 4
            issued = 2.0 * n * d * 32 * ((nl + 31) // 32) * ((2 if esz == 2 else 3) if split else 1)
            mpeak = MFMA16_PEAK_TFLOPS if split else MFMA32_PEAK_TFLOPS
            out[name] = {"ms": round(kern * 1e3, 3), "labels": nl,
                         "workload": f"{nl} text labels over {n} voxel rows x {d} {'fp32' if esz == 4 else 'bf16'}, " + ("softmax, last column" if last else "feature surgery, [N, L] out"),
                         "roofline": {"kernel": ("query_split16_kernel" if esz == 2 else "query_split_kernel") if split else "query_mfma_kernel", "bound": "hbm",
                                      "achieved": round(nbytes / kern / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(nbytes / kern / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                      "algorithmic_bytes_per_launch": int(nbytes),
                                      "floors_ms": {"hbm": round(nbytes / (HBM_PEAK_GBS * 1e9) * 1e3, 2),
                                                    (("fp16_mfma_x2" if esz == 2 else "fp16_mfma_x3") if split else "fp32_mfma"): round(issued / (mpeak * 1e12) * 1e3, 2)}}}
            if split:  # the same scan on the exact-fp32 matrix instructions (SAF_Q_SPLIT is read per call), for the record
                os.environ["SAF_Q_SPLIT"] = "0"
                try:
                    fn()
                    torch.cuda.synchronize()
                    e0.record()
                    fn()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    out[name]["exact_fp32_mfma_ms"] = round(e0.elapsed_time(e1) / 2, 3)
                    out[name]["note"] = ("fp32 scores from fp16 matrix instructions (operands cut into fp16 pieces under power-of-two scales, fp32 "
                                         "accumulation; error <= 3 x 2^-22 of sum |a b| per score: tests/test_split_scan.py); "
                                         "`exact_fp32_mfma_ms`: the same call under SAF_Q_SPLIT=0 (v_mfma_f32_32x32x2_f32)")
                finally:
                    del os.environ["SAF_Q_SPLIT"]
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r06", "split_scan_traffic.json")))
                if split and d == 512 and esz == 4:
                    out[name]["roofline"]["traffic"] = tj["L5_softmax_last" if nl == 5 else "L63_surgery"]["hbm_bytes_per_launch"]
                    out[name]["roofline"]["traffic_source"] = "profiles/r06/split_scan_traffic.json (this round, another box): " + tj["method"]
            except Exception:  # noqa: BLE001
                pass
            hold.clear()
        del f32, b16
        torch.cuda.empty_cache()

    big = torch.empty((n, (q + 63) // 64 * 64), dtype=torch.float16, device=device)[:, :q]  # (rows padded to whole 128-byte lines, as query_scan_wide allocates its own output)

    def scan_case(name, fn, n_q, out_bytes):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        fn()
        e1.record()
        torch.cuda.synchronize()
        kern = e0.elapsed_time(e1) * 1e-3 / 2
        flop = 2.0 * n * d * n_q
        out[name] = {"value": round(q / kern, 1), "unit": "queries/s", "ms": round(kern * 1e3, 3),
                     "workload": f"{q} fp16 queries over {n} voxel rows x {d} (BASELINE config 5)",
                     "roofline": {"kernel": "query_wide3_kernel", "bound": "mfma", "achieved": round(flop / kern / 1e12, 1),
                                  "peak": MFMA16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flop / kern / 1e12 / MFMA16_PEAK_TFLOPS, 4),
                                  "traffic": None, "algorithmic_bytes_per_launch": int(n * d * 2 + n_q * d * 2 + out_bytes)}}
        try:  # counter traffic of the same scan: measured with tools/r06_query_traffic.sh (two profiler passes over bench.py --query)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r06", "query_traffic.json")))
            if tj.get(name) and a.dim == 512:
                out[name]["roofline"]["traffic"] = tj[name]["hbm_bytes_per_launch"]
                out[name]["roofline"]["traffic_read_bytes"], out[name]["roofline"]["traffic_write_bytes"] = tj[name]["read"], tj[name]["write"]
                out[name]["roofline"]["traffic_source"] = "profiles/r06/query_traffic.json (this round, another box): " + tj["method"]
        except Exception:  # noqa: BLE001
            pass

    scan_case("config5_row_argmax", lambda: query_scan_wide(feats16, text[n_bg:], "row_argmax"), q, n * 8)
    scan_case("config5_heat_maps", lambda: query_scan_wide(feats16, text, "vs_background", scale=100.0, n_background=n_bg,
                                                             rescale=True, out=big), q + n_bg, n * q * 2)
    scan_case("config5_query_max", lambda: query_scan_wide(feats16, text[n_bg:], "query_max"), q, q * 12)
    del feats16, big
    torch.cuda.empty_cache()

    # ---- config 3 END TO END: kMaX (ConvNeXt-L-shaped, random weights) + CLIP (ViT-B/32-shaped) jointly fused into the
    #      256^3 bf16 volume with label histogram, one frame per integrate() call as the reference drives it
    try:
        from spatially_aware_ai_amd.backbones import RandomKmaxConvNeXtL, RandomViTB32
        from spatially_aware_ai_amd.clipfusion import Clip
        from spatially_aware_ai_amd.segmentation import KmaxSegmentationModel

        clip = Clip("ViT-B-32 (random weights)", None, backbone=RandomViTB32(), tokenizer=None).to(device).eval()
        clip.requires_grad_(False)
        kmax = RandomKmaxConvNeXtL().to(device).eval().to(memory_format=torch.channels_last)
        kmax.requires_grad_(False)
        seg = KmaxSegmentationModel(kmax, device)
        grid = syn.make_grid(256)
        fz = ClipSeemFusion(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, False, a.height // 3, a.height // 6, clip, seg,
                            keep_xyz_world=False, feat_dtype=torch.bfloat16).to(device)
        nfr = min(32, uniq)

        def run():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                for i in range(nfr):
                    fz.integrate(depth[i:i + 1], rgb[i:i + 1], poses[i:i + 1], ks[i:i + 1])
                fz.flush()

        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        d_all = time.perf_counter() - t0
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            t0 = time.perf_counter()
            for i in range(nfr):
                seg.run_on_image(rgb[i].permute(2, 0, 1))
            torch.cuda.synchronize()
            d_seg = time.perf_counter() - t0
        out["config3_end_to_end"] = {
            "value": round(nfr / d_all, 2), "unit": "frames/s", "frames": nfr, "kmax_only_frames_per_s": round(nfr / d_seg, 2),
            "workload": "ClipSeemFusion.integrate, one 640x480 frame per call: kMaX-DeepLab-shaped panoptic model (ConvNeXt-L "
                        "encoder at 1281x960, random weights, bf16 autocast) per call, the ViT-B/32-shaped CLIP tower deferred to "
                        "the queue's flush, 256^3 x 512 bf16 volume + label histogram",
            "roofline": None,
            "note": "bound by the PyTorch-ROCm ConvNeXt-L forward (about 0.84 TFLOP per frame; library convolutions / GEMMs), "
                    "not by the fused path"}
        del fz, clip, kmax, seg
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline down
        out["config3_end_to_end"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    try:
        out["scene_latency_s"] = bench_scene(a, device)
    except Exception as e:  # noqa: BLE001
        out["scene_latency_s"] = {"total": None, "error": f"{type(e).__name__}: {e}"[:300]}
    out["seconds"] = round(time.perf_counter() - t_begin, 1)
    return out


def bench_scene(a, device, n_frames=300, out_dir=None):
    """The reference's own order on ONE scan, timed stage by stage (spatially_aware_ai_amd/scene.py): backproject_pcd ->
    scene_bounds -> ClipSeemFusion.integrate one frame per call (host -> device copies of every frame included, as the
    manager's loop has them, clip_seem_fusion.py:305-313) -> label_index -> discover_objects -> the attributes the manager
    sets from outside -> extract_mesh's 6-tuple -> per-object meshes -> artefacts on disk -> clip_text_query("chair").
    The scan: `n_frames` 640 x 480 frames of the analytic sphere-in-a-box scene whose bounds at 2 cm voxels come out as the
    reference's largest recorded grid, 127 x 104 x 116 (voxel_grid_compare.md:1-23), D = 512; backbone outputs are replayed
    (resident maps), so this is the latency of everything BEHIND the backbones -- the counterpart of README.md:4's "within a
    few minutes after a user scans the environment"."""
    import shutil
    import tempfile

    from spatially_aware_ai_amd.scene import reconstruct_scene

    t_gen = time.perf_counter()
    n_thr = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, host_cores())))  # (the scan is generated on the host: a box that reports 256 CPUs and grants 16 crawls with 256 threads)
    names, colors = syn.scene_class_names(), syn.scene_class_colors()
    cfg = {"voxel_size": 0.02, "trunc_vox": 3, "clip_patch_size": a.height // 3, "clip_patch_stride": a.height // 6}
    warm = syn.SyntheticScan(3, 24, a.width, a.height, a.dim, box_half=syn.REFERENCE_GRID_BOX_HALF)
    tmp = out_dir or tempfile.mkdtemp(prefix="saf_scene_", dir="/tmp")
    try:
        # (every scan writes its own, new files -- the reference's paths carry the scan version, clip_seem_fusion.py:566-604;
        #  overwriting the warm-up's 3 GB file would add the truncation of its page cache: 0.8 instead of 0.35 s for the volume)
        r0 = reconstruct_scene(warm, cfg, syn.ReplayClip(warm, device, names), syn.ReplaySeg(warm, device), names, colors,
                               device=device, out_dir=os.path.join(tmp, "warm"))
        r0.text_query(syn.ReplayClip(warm, device, names), "chair")
        del r0
        torch.cuda.empty_cache()
        scan = syn.SyntheticScan(4, n_frames, a.width, a.height, a.dim, box_half=syn.REFERENCE_GRID_BOX_HALF)
        clip, seg = syn.ReplayClip(scan, device, names), syn.ReplaySeg(scan, device)
        t_gen = time.perf_counter() - t_gen
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = reconstruct_scene(scan, cfg, clip, seg, names, colors, device=device, out_dir=os.path.join(tmp, "scan"))
        t_rec = time.perf_counter() - t0
        ans = res.text_query(clip, "chair")
        t_q1 = res.seconds["text_query"]
        res.text_query(clip, "floor")  # a second query: the engine and its control set exist
        t_q2 = res.seconds["text_query"] - t_q1
        sizes = {k: os.path.getsize(v) for k, v in res.paths.items()}
        uo = res.scene_knowledge["unique_objects"]
        out = {
            "total": round(t_rec + t_q1, 3), "reconstruct": round(t_rec, 3), "first_text_query": round(t_q1, 3),
            "next_text_query": round(t_q2, 3), "stages": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.seconds.items() if k != "text_query"},
            "frames": n_frames, "image": [a.width, a.height], "grid": [int(v) for v in res.nvox], "feat_dim": a.dim,
            "fuse_frames_per_s": round(n_frames / res.seconds["fuse"], 1), "objects": len(uo),
            "object_labels": sorted({o["class_label"] for o in uo.values()}), "mesh_vertices": int(len(res.verts)),
            "mesh_faces": int(len(res.faces)), "artefact_bytes": int(sum(sizes.values())),
            "query_answer_colors": 0 if ans is None else len(ans["colors"]),
            "synthetic_scan_generation_s_untimed": round(t_gen, 1),
            "workload": f"{n_frames} frames {a.width}x{a.height} of the analytic scene (depth B, panoptic map = the surface's class, "
                        f"feature map = the class embedding + noise), one integrate() call per frame with the frame copied host -> "
                        f"device in the loop, {a.dim}-dim f32 features; backbones replayed",
            "reference": "clip_seem_fusion.py:247-437 (run_clipfusion) + :482-561 (clip_text_query); README.md:4 'within a few minutes'",
        }
        del res
        torch.cuda.empty_cache()
        return out
    finally:
        torch.set_num_threads(n_thr)
        if out_dir is None:
            shutil.rmtree(tmp, ignore_errors=True)


MFMA16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 matrix peak of MI355X (guides/MI355X_MICROARCH.md)
MFMA32_PEAK_TFLOPS = 157.3   # exact-fp32 matrix peak (= the fp32 vector rate)


def bench_query(a, world, rank, local_rank):
    """`bench.py --query`: the text-query scans over a volume resident in HBM, one JSON line in the same schema.

    The headline is BASELINE config 5 -- 1000 text queries over a 256^3 x 512 fp16 volume, the query_mesh.py path (4
    shared background prompts + one target per query, softmax(100 * cos)[-1] per voxel and target), voxel-sharded
    over the ranks: `value` = queries/s for the whole volume, writing the N x 1000 fp16 heat maps.  `cases` adds the
    fused reductions that write no N x Q matrix (per-voxel best query, per-query best voxel), the raw scores, and the
    reference's own small scans over the fp32 volume (L = 5 softmax: query_mesh.py:36-39; L = 63 surgery:
    query_mesh.py:52-83), each with its roofline from HIP events around the launch."""
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    if os.environ.get("SAF_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, **({"device_id": device} if a.backend == "nccl" else {}))
    from spatially_aware_ai_amd import distributed as sdist
    from spatially_aware_ai_amd.clipfusion import _query_scan, query_scan_wide

    n_all = a.grid3[0] * a.grid3[1] * a.grid3[2]
    first, n = sdist.voxel_shard(n_all, rank, world)
    d, q, n_bg = a.dim, a.queries, 4
    g = torch.Generator(device=device).manual_seed(100 + rank)
    feats16 = torch.empty((n, d), dtype=torch.float16, device=device)
    for s0 in range(0, n, 1 << 20):
        feats16[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), d), generator=g, device=device).half()
    tg = torch.Generator().manual_seed(9)
    text = torch.randn((n_bg + q, d), generator=tg)
    text = (text / text.norm(dim=-1, keepdim=True)).to(device)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for e0, e1 in ev:
            e0.record()
            fn()
            e1.record()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([wall], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall = float(tmax.item())
        return wall / steps, sum(e0.elapsed_time(e1) for e0, e1 in ev) / steps * 1e-3

    def mfma_case(name, fn, n_queries, out_bytes, steps, warmup):
        wall, kern = timed(fn, steps, warmup)
        flop = 2.0 * n * d * n_queries
        return {"case": name, "ms": round(wall * 1e3, 3), "rows_this_rank": n, "queries": n_queries,
                "roofline": {"bound": "mfma", "achieved": round(flop / kern / 1e12, 1), "peak": MFMA16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(flop / kern / 1e12 / MFMA16_PEAK_TFLOPS, 4), "traffic": None,
                             "avg_launch_us": round(kern * 1e6, 1),
                             "algorithmic_bytes_per_launch": int(n * d * 2 + n_queries * d * 2 + out_bytes),
                             "hbm_GBps_of_algorithmic_bytes": round((n * d * 2 + out_bytes) / kern / 1e9, 1)}}

    keep = {}
    qs = (q + 63) // 64 * 64  # rows padded to whole 128-byte lines (query_scan_wide allocates its own output the same way)
    big = torch.empty((n, qs), dtype=torch.float16, device=device)[:, :q]  # the N x Q output, allocated once

    def heat_maps():
        query_scan_wide(feats16, text, "vs_background", scale=100.0, n_background=n_bg, rescale=True, out=big)

    def query_max():
        val, row = query_scan_wide(feats16, text[n_bg:], "query_max", row_offset=first)
        if world > 1:  # the only exchange of the sharded scan: Q (score, voxel) pairs per rank
            vals = [torch.empty_like(val) for _ in range(world)]
            rows = [torch.empty_like(row) for _ in range(world)]
            dist.all_gather(vals, val)
            dist.all_gather(rows, row)
            keep["qm"] = torch.stack(vals).max(dim=0)
        else:
            keep["qm"] = (val, row)

    cases = []
    head = mfma_case("query_mesh path: softmax([4 backgrounds, target])[-1] heat maps for every target, fp16 out",
                     heat_maps, q + n_bg, n * q * 2, a.steps, a.warmup)
    cases.append(head)
    cases.append(mfma_case("per-voxel best query (row_argmax), no N x Q output",
                           lambda: query_scan_wide(feats16, text[n_bg:], "row_argmax"), q, n * 8, a.steps, 1))
    cases.append(mfma_case("per-query best voxel (query_max), no N x Q output", query_max, q, q * 12, a.steps, 1))
    cases.append(mfma_case("raw scores, fp16 out", lambda: query_scan_wide(feats16, text[n_bg:], "scores", out=big), q,
                           n * q * 2, a.steps, 1))
    keep.clear()
    del big
    torch.cuda.empty_cache()
    # the reference's own scans over the fp32 volume (single GPU only: they are HBM-bound and tiny next to the above)
    if world == 1 and not a.query_wide_only:
        del feats16
        torch.cuda.empty_cache()
        f32 = torch.empty((n, d), dtype=torch.float32, device=device)
        for s0 in range(0, n, 1 << 20):
            f32[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), d), generator=g, device=device)
        for name, nl, epi, scale, peak_note in (
                ("L=5 softmax over the fp32 volume (query_mesh.py:36-39), last column", 5, _abi.SAF_Q_SOFTMAX, 100.0, None),
                ("L=63 surgery over the fp32 volume (query_mesh.py:52-83)", 63, _abi.SAF_Q_SURGERY, 1.0, None)):
            t = text[:nl] if nl <= text.shape[0] else torch.nn.functional.normalize(torch.randn((nl, d), device=device), dim=-1)
            last = epi == _abi.SAF_Q_SOFTMAX
            wall, kern = timed(lambda: keep.__setitem__("o", _query_scan(f32, t, epi, scale=scale, normalize=True, last_only=last)),
                               a.steps, 1)
            nbytes = n * d * 4 + nl * d * 4 + n * (1 if last else nl) * 4
            flop = 2.0 * n * d * nl
            # what the kernel's matrix pipes execute: whole 32-label tiles (5 labels: one tile; 63: two).  The split scan (round 6,
            # the default): three fp16 MFMA products per fp32 product (hi.hi + hi.lo + lo.hi, fp32 accumulation) on the fp16 pipes;
            # SAF_Q_SPLIT=0: exact-fp32 MFMAs on the fp32 pipes.
            split = os.environ.get("SAF_Q_SPLIT", "1") != "0" and d % 16 == 0
            tiles_flop = 2.0 * n * d * 32 * ((nl + 31) // 32)
            flop_issued = tiles_flop * (3 if split else 1)
            mpeak = MFMA16_PEAK_TFLOPS if split else MFMA32_PEAK_TFLOPS
            t_hbm, t_mfma = nbytes / (HBM_PEAK_GBS * 1e9), flop_issued / (mpeak * 1e12)
            hbm = {"achieved": round(nbytes / kern / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / kern / 1e9 / HBM_PEAK_GBS, 4)}
            mf = {"achieved": round(flop_issued / kern / 1e12, 1), "peak": mpeak, "unit": "TFLOP/s",
                  "frac": round(flop_issued / kern / 1e12 / mpeak, 4)}
            bound = "mfma" if t_mfma > t_hbm else "hbm"  # the larger of the two floors names the bound
            traffic, traffic_source = None, None
            try:  # counter traffic of the same scan at the full size: tools/r06_split_traffic.sh (two profiler passes)
                tj = json.load(open(os.path.join(ROOT, "profiles", "r06", "split_scan_traffic.json")))
                if split and n == 1 << 24 and d == 512:
                    traffic = tj["L5_softmax_last" if nl == 5 else "L63_surgery"]["hbm_bytes_per_launch"]
                    traffic_source = "profiles/r06/split_scan_traffic.json (this round, another box): " + tj["method"]
            except (OSError, KeyError, ValueError):
                pass
            cases.append({"case": name, "ms": round(wall * 1e3, 3), "rows_this_rank": n, "queries": nl,
                          "roofline": dict(mf if bound == "mfma" else hbm, bound=bound, traffic=traffic, traffic_source=traffic_source,
                                           avg_launch_us=round(kern * 1e6, 1),
                                           kernel="query_split_kernel" if split else "query_mfma_kernel",
                                           algorithmic_bytes_per_launch=int(nbytes), issued_mfma_flop_per_launch=int(flop_issued),
                                           floors_ms={"hbm": round(t_hbm * 1e3, 2), ("fp16_mfma_x3" if split else "fp32_mfma"): round(t_mfma * 1e3, 2)},
                                           other_bound=(hbm if bound == "mfma" else mf),
                                           fp32_product_TFLOPs=round(flop / kern / 1e12, 1),
                                           note=("fp32 scores from fp16 matrix instructions: every fp32 operand cut into two fp16 pieces under a "
                                                 "power-of-two scale that follows the row's running maximum, hi.hi + hi.lo + lo.hi accumulated in "
                                                 "fp32 (error <= 3 x 2^-22 of sum |a b| per score, tests/test_split_scan.py) + one pass over the "
                                                 "fp32 volume: both floors are stated; `frac` is against the larger one"
                                                 if split else
                                                 "exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) + one pass over the fp32 volume: both "
                                                 "floors are stated; `frac` is against the larger one"))})
            keep.clear()
    if rank == 0:
        out = {
            "metric": f"text queries/s over a {a.grid}^3x{d} fp16 volume (BASELINE config 5: 1000-query CLIP-text scan, query_mesh path)",
            "value": round(q / (head["ms"] * 1e-3), 1), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"{q} target queries + {n_bg} shared background prompts over {n_all} voxel rows x {d} fp16, "
                                   f"voxel-sharded over {world} rank(s); a step = one scan of the whole volume writing the "
                                   f"[N, {q}] fp16 heat maps", "grid": a.grid, "feat_dim": d, "queries": q, "n_voxels": n_all,
                       "parallelism": f"voxels-shard{world}", "rccl_world": world, "backend": a.backend if world > 1 else None},
            "roofline": dict(head["roofline"], kernel="query_wide3_kernel"),
            "cases": cases,
            "cpu_baseline": query_cpu_baseline(a, feats_shape=(n_all, d), q=q + n_bg) if world == 1 and a.cpu_frames != 0 else None,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def query_cpu_baseline(a, feats_shape, q):
    """The CPU oracle's scan (saf_oracle_query_scan: scores in double, the checker of the parity tests) on a bounded
    sample of rows of the same shape, one thread (the restatement is scalar C)."""
    try:
        from oracle import oracle as O
    except Exception as e:
        return {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    rows = 2048
    g = torch.Generator().manual_seed(3)
    f = torch.randn(rows, feats_shape[1], generator=g).half().float()
    t = torch.nn.functional.normalize(torch.randn(q, feats_shape[1], generator=g), dim=-1)
    t0 = time.perf_counter()
    O.wide_scan(f, t, "vs_background", scale=100.0, n_background=4)
    dt = time.perf_counter() - t0
    # queries/s for the whole volume at this rate
    return {"value": round((q - 4) / (dt * feats_shape[0] / rows), 4), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"{rows} of the {feats_shape[0]} rows x {q} text rows in {dt:.2f} s on one host thread, "
                      "extrapolated to the whole volume", "cpu_model": cpu_model()}


def bench_api_b1(a, fusion, depth, rgb, poses, ks, feat, label_maps, bulk_value):
    """The reference's own call pattern (clipfusion.py:1125-1133, clip_seem_fusion.py:303-313): ONE frame per
    integrate() call, here through integrate_features() (backbone outputs resident).  The deferred window queue
    behind it (clipfusion._FusionVolumeMixin._fuse) copies each call's inputs into a staging ring and fuses 128
    frames at a time on the windowed path; a job = reset + n calls + the final flush, like a bulk step."""
    n, nu = a.api_b1, depth.shape[0]  # (more calls than resident frames: the frames again, in order)

    def job():
        fusion.reset()
        for k in range(n):
            i = k % nu
            labs = None if label_maps is None else [label_maps[i]]
            fusion.integrate_features(depth[i:i + 1], rgb[i:i + 1], poses[i:i + 1], ks[i:i + 1], feat[i:i + 1], labs)
        fusion.flush()

    job()
    torch.cuda.synchronize()
    if os.environ.get("SAF_BENCH_PROFILE_API") == "1":  # development: which calls of the loop block the host?
        ts = []
        fusion.reset()
        for k in range(n):
            i = k % nu
            t = time.perf_counter()
            fusion.integrate_features(depth[i:i + 1], rgb[i:i + 1], poses[i:i + 1], ks[i:i + 1], feat[i:i + 1],
                                      None if label_maps is None else [label_maps[i]])
            ts.append(time.perf_counter() - t)
        t = time.perf_counter()
        fusion.flush()
        ts.append(time.perf_counter() - t)
        torch.cuda.synchronize()
        print("api_b1 host us per call, buckets of 32:", " ".join(f"{sum(ts[i:i + 32]) / 32 * 1e6:.0f}" for i in range(0, n, 32)),
              " slowest (us, call):", sorted(((round(t * 1e6), i) for i, t in enumerate(ts)), reverse=True)[:6], file=sys.stderr)
    fusion.fuse_stats.zero_()
    import gc

    gc.collect()  # a full collection of this process takes ~40 ms: not inside the timed loop (timeit's hygiene)
    t0 = time.perf_counter()
    job()
    host_s = time.perf_counter() - t0  # the host has queued everything
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = fusion.stats()
    assert st["frames"] == n and int(fusion.weight.sum(dtype=torch.int64)) == st["valid"]
    return {"value": round(n / dt, 2), "unit": "frames/s", "frames": n, "frames_per_call": 1,
            "vs_bulk": round(n / dt / bulk_value, 4), "host_enqueue_us_per_call": round(host_s / n * 1e6, 1),
            "windowed": st["window_rows"] > 0,
            "note": "one frame per integrate_features() call, deferred window queue (%d-frame windows)" % WIN + ", includes the reset "
                    "of the volume and the final flush; `vs_bulk` = this / the bulk `value` of the same run"}


def bench_voxel_sharded(a, dist, sdist, grid, fusions, tensors, new_volume, world, rank, device, L, stream, npy, npx):
    """The same job with VOXELS sharded instead of frames: rank k owns a slab of x-planes of the
    volume (blocks of 16 planes, an outer block paired with an inner one: cameras look at the middle of the scene) and
    fuses ALL world * frames_per_rank frames into it.  The only exchange is the frames -- 5 MB per frame instead of the
    34 GB of a volume merge -- all-gathered SEGMENT BY SEGMENT on a side stream while the previous segment is being fused
    (every rank starts with its own frames only; the exchange is inside the timed step); there is no merge, and the job
    ends voxel-sharded.  Frame order of the job: segment-major, ranks ascending within a segment.  Untimed check: the slab
    equals, bit for bit, the same x-planes of a full-size volume fused on this rank alone from the same frames in the same
    order."""
    nx, ny, nz = (int(v) for v in grid.nvox)
    planes = sdist.slab_planes_of_rank(nx, rank, world)
    cnt = int(planes.numel())
    slab = new_volume(torch.tensor([cnt, ny, nz], dtype=torch.int32), x_planes=planes)
    ws = slab._get_workspace(npy, npx, (a.height, a.width))
    stats = slab._buffers["fuse_stats"]
    uniq = tensors[0].shape[0]
    # segments of `seg` frames per rank: at least two windows per fuse call, so that classification and rows overlap inside it
    seg = max(64, 2 * WIN // world)
    while uniq % seg != 0 or a.frames % seg != 0:
        seg //= 2
        if seg < 16:
            seg = uniq  # one segment: the exchange in front of the fusion
            break
    n_seg = uniq // seg
    recv = [[None if t is None else torch.empty((world * seg,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in tensors]
            for _ in range(n_seg)]
    descs = []
    for r in recv:
        arr, keep, _, _ = slab._make_frames(*r[:5], r[5], a.labels)
        descs.append((arr, keep))
    calls = [k % n_seg for k in range(a.frames // seg)]  # a.frames > uniq: the unique frames again (their buffers are resident)
    comm = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    nccl = dist.get_backend() == "nccl"

    def exchange(k):
        for t, dst in zip(tensors, recv[k]):
            if t is None:
                continue
            src = t[k * seg:(k + 1) * seg].contiguous()
            if nccl:
                dist.all_gather_into_tensor(dst, src)
            else:
                dist.all_gather(list(dst.split(seg)), src)

    def job():
        slab.reset()
        free = main.record_event()  # the receive buffers are free once the previous job's fuse calls are done
        comm.wait_event(free)
        arrived = []
        with torch.cuda.stream(comm):
            for k in range(n_seg):
                exchange(k)
                arrived.append(comm.record_event())
        vol = slab._c_volume(for_fuse=True)
        for i, k in enumerate(calls):
            if i < n_seg:
                main.wait_event(arrived[k])
            check(L.saf_fuse_frames_profiled(C.byref(vol), descs[k][0], world * seg, ws.data_ptr(), ws.numel(), stats.data_ptr(), None,
                                             stream), "saf_fuse_frames (slab)")
        slab.flush()

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    job()
    barrier()
    stats.zero_()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        job()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    st = slab.stats()
    assert st["frames"] == world * a.frames * a.steps
    assert int(slab.weight.sum(dtype=torch.int64)) * a.steps == st["valid"]
    # untimed: a full-size volume fused here from a few of the same frames must agree with the slab on its x-planes
    c = min(max(1, a.check_frames), seg)
    sub = (_abi.SafFrame * (world * c))()
    for r in range(world):
        for i in range(c):
            sub[r * c + i] = descs[0][0][r * seg + i]
    full = fusions[1]
    full.reset(lazy=False)  # (a handful of frames: the per-frame pipeline reads the rows it updates)
    wsf = full._get_workspace(npy, npx, (a.height, a.width))
    volf = full._c_volume(for_fuse=True)
    check(L.saf_fuse_frames_profiled(C.byref(volf), sub, world * c, wsf.data_ptr(), wsf.numel(), full._buffers["fuse_stats"].data_ptr(),
                                     None, stream), "check (full volume)")
    slab.reset(lazy=False)
    vols = slab._c_volume(for_fuse=True)
    check(L.saf_fuse_frames_profiled(C.byref(vols), sub, world * c, ws.data_ptr(), ws.numel(), stats.data_ptr(), None, stream),
          "check (slab)")
    pl = planes.to(device)
    mine_of = lambda t: t.view(nx, ny * nz, -1).index_select(0, pl).reshape(cnt * ny * nz, -1)
    same = all(bool(torch.equal(getattr(slab, n).view(cnt * ny * nz, -1), mine_of(getattr(full, n))))
               for n in ("weight", "tsdf_weight", "tsdf", "rgb", "clip_feat"))
    flag = torch.tensor([int(same), int((slab.weight > 0).sum())], device=device, dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    assert int(flag[0]) == 1, "voxel-sharded check: a slab differs from the same x-planes of the full-size volume"
    assert int(flag[1]) > 0, "voxel-sharded check: a slab was not touched by the check frames"
    del slab, recv, descs
    torch.cuda.empty_cache()
    total = world * a.frames * a.steps
    return {"value": round(total / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / a.steps * 1e3, 3),
            "slab_voxels": cnt * ny * nz, "x_planes": "blocks of 16, outer paired with inner" if cnt and int(planes[-1] - planes[0]) + 1 != cnt else "contiguous",
            "frames_fused_per_rank_per_step": world * a.frames, "collective": "all_gather of the frames",
            "segments": n_seg, "frames_per_rank_per_segment": seg,
            "slab_equals_full_volume_range": True,
            "note": "every rank fuses all frames of the job into its slab of x-planes: no merge; the frames are all-gathered "
                    "segment by segment on a side stream beside the fusion of the previous segment (inside the timed step); "
                    "the volume stays voxel-sharded; NOT BASELINE config 4's layout (that is `value`)"}


def check_merge(a, dist, sdist, fusions, fuse_into, merge, frames, tensors, c, world, rank, device, pipelined=None):
    """Untimed proof of the merge on the full-size volumes.  Every rank fuses its first c frames in SAF_SUM mode and
    the ranks merge (the very collective that was timed); every rank then all-gathers the c frames of all ranks and
    fuses ALL world*c frames, rank after rank, into its second volume as running means -- what a single GPU would
    have produced.  The two must agree on the voxel stripes this rank owns: integer weights exactly, means to 1e-4.
    ``pipelined(fz, frames, c) -> stripes``: the slab-pipelined form of the same job is proved the same way (its stripes,
    its finalize on the communication stream) -- a region may only become the headline after its VALUES were compared."""
    v0, v1 = fusions[0], fusions[1]
    v0.reset(accum_mode=_abi.SAF_SUM)
    fuse_into(v0, frames, c, None)
    stripes = merge(v0)
    # every rank's check frames, gathered (c x ~5 MB per rank)
    gathered = []
    for t in tensors:
        if t is None:
            gathered.append(None)
            continue
        mine = t[:c].contiguous()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        gathered.append(torch.cat(parts))
    depth, rgb, poses, ks, feat, labs = gathered
    arr, keep, _, _ = v1._make_frames(depth, rgb, poses, ks, feat, labs, a.labels)
    v1.reset(accum_mode=_abi.SAF_RUNNING_MEAN)
    fuse_into(v1, arr, world * c, None)
    v1.flush()
    torch.cuda.synchronize()

    def compare(stripes, what):
        ok_w, worst, touched = True, 0.0, 0
        b0, b1 = v0._buffers, v1._buffers  # (raw buffers: v0 holds stripes only and must not be "looked at" as a whole)
        for first, count in stripes:
            sl = slice(first, first + count)
            ok_w = ok_w and bool(torch.equal(b0["weight"][sl], b1["weight"][sl])) and bool(torch.equal(b0["tsdf_weight"][sl], b1["tsdf_weight"][sl]))
            if a.labels:
                ok_w = ok_w and bool(torch.equal(b0["labels_one_hot"][sl], b1["labels_one_hot"][sl]))
            for name in ("clip_feat", "rgb", "tsdf"):
                x, y = b0[name][sl], b1[name][sl]
                step = max(1, (1 << 22) // max(1, x[0].numel()))  # compare in pieces: no full-size temporaries
                for s0 in range(0, x.shape[0], step):
                    xs, ys = x[s0:s0 + step].float(), y[s0:s0 + step].float()
                    worst = max(worst, float(((xs - ys).abs() / (ys.abs() * 1e-4 + 1e-6)).max()))
            touched += int((b1["weight"][sl] > 0).sum())
        res = torch.tensor([int(ok_w), int(worst <= 1.0), touched], device=device, dtype=torch.int64)
        dist.all_reduce(res, op=dist.ReduceOp.MIN)
        all_ok_w, all_ok_f, min_touched = (int(v) for v in res.tolist())
        assert all_ok_w, f"merge check ({what}): merged integer weights differ from the single-rank fusion of the same frames"
        assert all_ok_f, f"merge check ({what}): merged means differ from the single-rank fusion beyond 1e-4 (worst ratio {worst:.3g})"
        assert min_touched > 0, f"merge check ({what}): a rank's voxel stripes were not touched by the check frames"
        cover = torch.tensor([sum(c_ for _, c_ in stripes)], device=device, dtype=torch.int64)
        dist.all_reduce(cover)
        return {"stripes_of_this_rank": len(stripes), "owned_voxels_all_ranks": int(cover), "weights_exact": True,
                "means_within_1e-4": True, "worst_error_over_tolerance": round(worst, 4), "touched_voxels_min_over_ranks": min_touched}

    out = {"check_frames_per_rank": c, "serial_merge_values": compare(stripes, "serial merge")}
    if pipelined is not None:
        v0.reset(accum_mode=_abi.SAF_SUM)
        out["slab_pipelined_values"] = compare(pipelined(v0, frames, c), "slab-pipelined merge")
    return out


def host_cores():
    """Host threads this process may really use: the scheduler affinity capped by the cgroup CPU
    quota (the GPU boxes expose 256 hardware threads but grant a 16-CPU share per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _time_oracle(O, vol, tensors, n, budget_s):
    d, r, p, k, f = tensors
    vol.integrate(d[:1], r[:1], p[:1], k[:1], f[:1])  # touch pages / warm up
    t0 = time.perf_counter()
    done = 0
    for i in range(1, n):
        vol.integrate(d[i : i + 1], r[i : i + 1], p[i : i + 1], k[i : i + 1], f[i : i + 1])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return done, time.perf_counter() - t0


def cpu_baseline(a, grid, depth, rgb, poses, ks, feat, npy, npx):
    """Times oracle/saf_oracle.c (test infrastructure, used here only as the reported CPU baseline) on the first
    few frames of this workload: all host cores of the box's share, then one thread, then BASELINE config 1
    (32 frames 320x240 into 64^3 x 64) in full -- the plan of BASELINE.md section 3.  ~25 s in all."""
    try:
        from oracle import oracle as O
    except Exception as e:  # the oracle is optional for the benchmark line
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    cores = host_cores()
    n = a.cpu_frames if a.cpu_frames > 0 else 129
    n = min(n, depth.shape[0])
    tensors = tuple(t[:n].cpu() for t in (depth, rgb, poses, ks, feat))
    O.set_threads(cores)
    vol = O.OracleVolume(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, a.dim)
    done, dt = _time_oracle(O, vol, tensors, n, 10.0)
    O.set_threads(1)
    vol1 = O.OracleVolume(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, a.dim)
    done1, dt1 = _time_oracle(O, vol1, tensors, min(n, 9), 8.0)
    del vol, vol1
    # BASELINE config 1 in full: 32 synthetic 320x240 frames, 64^3 grid, 64-dim features (the reference's own
    # CPU-runnable case; its genuine Python path measured 39-49 frames/s on 8 threads and 13 on one)
    c1 = {}
    g1 = syn.make_grid(64)
    y1, x1 = syn.feature_map_shape(320, 240)
    fr1 = syn.make_frames(2024, 32, width=320, height=240, feat_dim=64, npy=y1, npx=x1, depth_kind="A")
    t1 = tuple(torch.cat([f[k] for f in fr1]) for k in ("depth", "rgb", "pose", "K", "feat"))
    for threads in (cores, 1):
        O.set_threads(threads)
        v = O.OracleVolume(g1.origin, g1.voxel_size, g1.nvox, g1.trunc, 64)
        dn, dtt = _time_oracle(O, v, t1, 32, 10.0)
        c1[f"threads_{threads}"] = round(dn / dtt, 2) if dn else None
    O.set_threads(1)
    return {
        "value": round(done / dt, 3) if done else None, "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{done} frames of the same workload ({a.grid}^3x{a.dim}, {a.width}x{a.height}), "
                  f"oracle/saf_oracle.c with OpenMP over {cores} host threads",
        "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(),
        "one_thread": {"value": round(done1 / dt1, 3) if done1 else None, "frames": done1},
        "config1_full": {"frames_per_s": c1, "sample": "BASELINE config 1 in full: 32 frames 320x240 into 64^3 x 64 (31 timed after one warm-up frame)"},
    }


if __name__ == "__main__":
    main()
