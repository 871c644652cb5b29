"""CPU, world_size 2, gloo: the frame-sharded job of SURVEY.md §8e.

Each rank fuses its contiguous block of frames into a private SUM-mode volume (with the CPU
oracle standing in for the HIP kernels -- this test covers the sharding + merge logic of
spatially_aware_ai_amd.distributed, which is device-agnostic), the ranks merge with one
collective, and the result must equal the single-process running-mean volume over all frames:
index sets / integer tensors exactly, fp32 within 1e-4 relative."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spatially_aware_ai_amd import _abi
from spatially_aware_ai_amd import distributed as sdist
from spatially_aware_ai_amd import synthetic as syn

NVOX = (9, 7, 5)  # 315 voxels: not divisible by the world size -> exercises the remainder path
DIM, W, H = 8, 40, 30
N_FRAMES = 7


def _frames(depth_kind="A"):
    npy, npx = syn.feature_map_shape(W, H)
    return syn.make_frames(321, N_FRAMES, width=W, height=H, feat_dim=DIM, npy=npy, npx=npx, depth_kind=depth_kind), (npy, npx)


def _fuse(vol, frames, seem):
    for f in frames:
        vol.integrate(f["depth"], f["rgb"], f["pose"], f["K"], f["feat"],
                      [f["labels"].float()] if seem else None, rgb_bilinear=seem)


def _tensors(vol):
    t = {"clip_feat": vol.clip_feat, "rgb": vol.rgb, "tsdf": vol.tsdf, "weight": vol.weight,
         "tsdf_weight": vol.tsdf_weight}
    if vol.labels_one_hot is not None:
        t["labels_one_hot"] = vol.labels_one_hot
    return t


def _worker(rank, world, port, mode, gather, out_dir, piece_bytes=None, sparse=None, nvox=NVOX, side=1.2, depth_kind="A",
            trunc_vox=3.0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O

        grid = syn.make_grid(nvox, side=side, trunc_vox=trunc_vox)
        frames, _ = _frames(depth_kind)
        mine = sdist.shard_frames(len(frames), rank, world)
        vol = O.OracleVolume(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, DIM, 143, _abi.SAF_SUM)
        _fuse(vol, [frames[i] for i in mine], seem=True)
        if sparse == "env_mismatch":  # ranks whose environments differ must still pick the SAME route: the minimum of their wishes
            os.environ["SAF_MERGE_SPARSE"] = "0" if rank == 0 else "1.0"
            sparse = None
        elif sparse == "library_default":  # nobody probed, nobody set a threshold: the library probes once, then 0.5
            os.environ.pop("SAF_MERGE_SPARSE", None)
            sparse = None
        elif sparse is not None:
            assert sdist.probe_all_to_all(torch.device("cpu")) is None
        stripes = sdist.merge_sums(_tensors(vol), mode=mode, gather=gather, piece_bytes=piece_bytes, sparse=sparse)
        if rank == 0:
            np.save(os.path.join(out_dir, "last_merge.npy"), np.array([sdist.last_merge[k] for k in ("pieces", "packed", "rows", "touched_rows")]))
        # local divide over the stripes this rank owns (the HIP path calls saf_merge_finalize here)
        c = vol.c_volume()
        import ctypes as C

        for first, count in stripes:
            assert O.lib().saf_oracle_merge_finalize(C.byref(c), first, count) == 0
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), stripes=np.asarray(stripes, dtype=np.int64).reshape(-1, 2),
                 **{k: v.numpy() for k, v in _tensors(vol).items()})
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_frames_partitions():
    for n, w in ((7, 2), (512, 8), (5, 8), (4096, 8)):
        blocks = [sdist.shard_frames(n, r, w) for r in range(w)]
        assert sum(len(b) for b in blocks) == n
        flat = [i for b in blocks for i in b]
        assert flat == list(range(n))
    assert len(sdist.shard_frames(4096, 3, 8)) == 512
    assert sdist.voxel_shard(315, 0, 2) == (0, 157) and sdist.voxel_shard(315, 1, 2) == (157, 158)


@pytest.mark.parametrize("mode,world,piece_bytes", [("reduce_scatter", 2, None), ("all_reduce", 2, None),
                                                    # several pieces per tensor, a ragged last piece, a tail of < world rows:
                                                    # the layout arithmetic of the striped in-place collectives at world 2, 4, 8
                                                    ("reduce_scatter", 2, 143 * 4 * 20), ("reduce_scatter", 4, 143 * 4 * 24),
                                                    ("reduce_scatter", 8, 143 * 4 * 32)])
def test_merge_equals_single_process(tmp_path, oracle, mode, world, piece_bytes):
    mp.spawn(_worker, args=(world, _free_port(), mode, False, str(tmp_path), piece_bytes), nprocs=world, join=True)
    grid = syn.make_grid(NVOX, side=1.2)
    frames, _ = _frames()
    ref = oracle.OracleVolume(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, DIM, 143)
    _fuse(ref, frames, seem=True)
    n = ref.n
    covered = np.zeros(n, dtype=np.int32)
    for r in range(world):
        g = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        stripes = [tuple(int(v) for v in st) for st in g["stripes"]]
        if mode == "all_reduce":
            assert stripes == [(0, n)]
        elif piece_bytes is None:  # one piece: the k-th of world equal parts, the remainder on the last rank
            assert stripes == [sdist.voxel_shard(n, r, world)]
        else:
            plan = sdist.stripe_plan(n, world, sdist.piece_rows_for(143 * 4, world, piece_bytes))
            assert len(plan) >= 3 and stripes == sdist.stripes_of_rank(plan, r, world)
        for first, count in stripes:
            sl = slice(first, first + count)
            covered[sl] += 1
            assert np.array_equal(g["weight"][sl], ref.weight.numpy()[sl]), "merged valid counts differ"
            assert np.array_equal(g["tsdf_weight"][sl], ref.tsdf_weight.numpy()[sl])
            assert np.array_equal(g["labels_one_hot"][sl], ref.labels_one_hot.numpy()[sl])
            np.testing.assert_allclose(g["clip_feat"][sl], ref.clip_feat.numpy()[sl], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(g["rgb"][sl], ref.rgb.numpy()[sl], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(g["tsdf"][sl], ref.tsdf.numpy()[sl], rtol=1e-4, atol=2e-6)
    expect = world if mode == "all_reduce" else 1
    assert (covered == expect).all(), "every voxel must be finalised by exactly the ranks that own it"
    assert int(ref.weight.sum()) > 0


@pytest.mark.parametrize("world,sparse,want_packed", [(2, 1.0, "all"), (4, 1.0, "all"), (8, 1.0, "all"), (4, 0.3, "some"), (2, 0.0, "none"),
                                                      (4, "env_mismatch", "none"), (4, "library_default", "any")])
def test_sparse_merge_equals_single_process(tmp_path, oracle, world, sparse, want_packed):
    """The merge that skips what no rank touched ("exploit sparsity"): the coherent scene in a grid wider
    than the scene -- whole pieces of the stripe plan are empty, others thin shells -- merged with the touched rows packed
    (all_to_all with uneven splits) where the touched share of a piece is at most `sparse`, dense elsewhere: the same
    stripes, the same volume as one process fusing every frame."""
    nvox, side, piece_bytes = (16, 12, 10), 4.2, 143 * 4 * 48
    mp.spawn(_worker, args=(world, _free_port(), "reduce_scatter", False, str(tmp_path), piece_bytes, sparse, nvox, side, "B", 1.0),
             nprocs=world, join=True)
    grid = syn.make_grid(nvox, side=side, trunc_vox=1.0)
    frames, _ = _frames("B")
    ref = oracle.OracleVolume(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, DIM, 143)
    _fuse(ref, frames, seem=True)
    n = ref.n
    pieces, packed, rows, touched = np.load(os.path.join(tmp_path, "last_merge.npy")).tolist()
    plan = sdist.stripe_plan(n, world, sdist.piece_rows_for(143 * 4, world, piece_bytes))
    assert pieces == len(plan) >= 3 and rows == n
    w_ref = ref.weight.numpy()
    if want_packed == "none":
        assert packed == 0  # (rank 0 wished 0: it never looked at the touched rows)
    elif want_packed == "any":
        assert packed >= 1 and touched == int((w_ref > 0).sum())
    else:
        assert touched == int((w_ref > 0).sum()) and 0 < touched < 0.5 * n
        empty = sum(1 for first, r, c in plan if not (w_ref[first:first + world * c] > 0).any())
        assert empty >= 1, "the workload has whole pieces no rank touched"
        assert packed == pieces if want_packed == "all" else 0 < packed < pieces
    covered = np.zeros(n, dtype=np.int32)
    for r in range(world):
        g = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        stripes = [tuple(int(v) for v in st) for st in g["stripes"]]
        assert stripes == sdist.stripes_of_rank(plan, r, world)
        assert np.array_equal(g["weight"], w_ref), "weight is the job's total on EVERY row of every rank, whatever the route"
        for first, count in stripes:
            sl = slice(first, first + count)
            covered[sl] += 1
            assert np.array_equal(g["weight"][sl], w_ref[sl]) and np.array_equal(g["tsdf_weight"][sl], ref.tsdf_weight.numpy()[sl])
            assert np.array_equal(g["labels_one_hot"][sl], ref.labels_one_hot.numpy()[sl])
            np.testing.assert_allclose(g["clip_feat"][sl], ref.clip_feat.numpy()[sl], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(g["rgb"][sl], ref.rgb.numpy()[sl], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(g["tsdf"][sl], ref.tsdf.numpy()[sl], rtol=1e-4, atol=2e-6)
    assert (covered == 1).all()


# ---- BASELINE config 5: the voxel-sharded query scan (sharding + reductions; the scan itself is injected) ----
Q_TEXT = 21


def _query_worker(rank, world, port, out_dir, from_merge):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O

        grid = syn.make_grid(NVOX, side=1.2)
        frames, _ = _frames()
        vol = O.OracleVolume(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, DIM)
        _fuse(vol, frames, seem=False)
        if from_merge == "striped":  # the stripes a merge in several pieces leaves (5 pieces of 64 rows, a ragged last one)
            vol._shard_stripes = sdist.stripes_of_rank(sdist.stripe_plan(vol.n, world, 32 * world), rank, world)
        else:
            vol._shard_stripes = [sdist.voxel_shard(vol.n, rank, world)] if from_merge else None
        text = torch.randn(Q_TEXT, DIM, generator=torch.Generator().manual_seed(5))
        text[3] = text[11]  # two queries with identical scores everywhere
        res = {}
        for epi, kw in (("query_max", {}), ("row_argmax", {}), ("vs_background", {"n_background": 4, "scale": 100.0})):
            out = sdist.query_sharded(vol, text, epi, scan_fn=O.wide_scan, **kw)
            out = out if isinstance(out, tuple) else (out,)
            for i, t in enumerate(out):
                res[f"{epi}_{i}"] = t.numpy()
        np.savez(os.path.join(out_dir, f"q{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("from_merge,world", [(False, 2), (True, 2), ("striped", 2), ("striped", 4)])
def test_sharded_query_equals_single_process(tmp_path, oracle, from_merge, world):
    mp.spawn(_query_worker, args=(world, _free_port(), str(tmp_path), from_merge), nprocs=world, join=True)
    grid = syn.make_grid(NVOX, side=1.2)
    frames, _ = _frames()
    ref = oracle.OracleVolume(grid.origin, grid.voxel_size, grid.nvox, grid.trunc, DIM)
    _fuse(ref, frames, seem=False)
    text = torch.randn(Q_TEXT, DIM, generator=torch.Generator().manual_seed(5))
    text[3] = text[11]
    qv, qr = oracle.wide_scan(ref.clip_feat, text, "query_max")
    ri, rv = oracle.wide_scan(ref.clip_feat, text, "row_argmax")
    vb = oracle.wide_scan(ref.clip_feat, text, "vs_background", n_background=4, scale=100.0)
    assert int((ref.weight > 0).sum()) > 20
    for r in range(world):
        g = np.load(os.path.join(tmp_path, f"q{r}.npz"))
        # every rank ends with the same per-query answer as one scan over the whole volume (ties -> smaller voxel)
        assert np.array_equal(g["query_max_0"], qv.numpy()) and np.array_equal(g["query_max_1"], qr.numpy())
        assert np.array_equal(g["row_argmax_0"], ri.numpy()) and np.array_equal(g["row_argmax_1"], rv.numpy())
        if from_merge == "striped":
            stripes = sdist.stripes_of_rank(sdist.stripe_plan(ref.n, world, 32 * world), r, world)
            assert len(stripes) >= 3
        else:
            stripes = [sdist.voxel_shard(ref.n, r, world)]
        want = np.concatenate([vb.numpy()[f:f + c] for f, c in stripes])
        np.testing.assert_allclose(g["vs_background_0"], want, rtol=1e-6, atol=1e-7)


def _gather_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = [torch.full((3, 4, 5), float(rank)), None, torch.arange(3 * 2).view(3, 2) + 100 * rank]
        out = sdist.gather_frames(mine)
        assert out[1] is None
        np.savez(os.path.join(out_dir, f"g{rank}.npz"), a=out[0].numpy(), c=out[2].numpy())
    finally:
        dist.destroy_process_group()


def test_voxel_sharded_job_exchanges_frames_in_rank_order(tmp_path):
    """The voxel-sharded job's only exchange: every rank ends with all frames, rank after rank; slabs tile the x axis."""
    world = 2
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        g = np.load(os.path.join(tmp_path, f"g{r}.npz"))
        assert g["a"].shape == (6, 4, 5) and (g["a"][:3] == 0).all() and (g["a"][3:] == 1).all()
        assert g["c"][:, 0].tolist() == [0, 2, 4, 100, 102, 104]
    for nx, w in ((256, 8), (33, 3), (5, 8)):
        slabs = [sdist.slab_of_rank(nx, r, w) for r in range(w)]
        assert sum(c for _, c in slabs) == nx and all(slabs[i][0] + slabs[i][1] == slabs[i + 1][0] for i in range(w - 1))


def test_balanced_planes_partition_the_grid():
    """slab_planes_of_rank: every x-plane belongs to exactly one rank, ranks hold equally many, in whole blocks of 16, and
    a rank's blocks mirror each other (as far out on one side as far in on the other); odd shapes fall back to slabs."""
    from spatially_aware_ai_amd import distributed as sdist

    for nx, world in ((256, 1), (256, 2), (256, 4), (256, 8), (64, 2), (128, 4)):
        parts = [sdist.slab_planes_of_rank(nx, r, world) for r in range(world)]
        assert sorted(torch.cat(parts).tolist()) == list(range(nx))
        assert len({p.numel() for p in parts}) == 1
        for p in parts:
            assert torch.equal(p, torch.sort(p).values) and all(int(v) % 16 == 0 for v in p[::16])
            centre = (nx - 1) / 2
            assert abs(float((p.double() - centre).mean())) < 16 * world, "blocks of a rank do not mirror each other"
    assert sdist.slab_planes_of_rank(33, 1, 3).tolist() == list(range(11, 22))  # no blocks of 16: the contiguous slab


def _slab_merge_worker(rank, world, port, out, piece_bytes=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_rows = 7 * 5 * 3  # 7 x-planes of 15 voxels: slabs of unequal size, parts with a remainder
        g = torch.Generator().manual_seed(100 + rank)
        mine = {"clip_feat": torch.randn(n_rows, 6, generator=g), "weight": torch.randint(0, 9, (n_rows,), generator=g, dtype=torch.int32)}
        total = {}
        for k, v in mine.items():
            parts = [torch.empty_like(v) for _ in range(world)]
            dist.all_gather(parts, v)
            total[k] = torch.stack(parts).sum(0)
        why = sdist.probe_collectives(torch.device("cpu"))
        assert why is None, why
        stripes = []
        for x0, cnt in sdist.slab_bounds(7, 3):
            stripes += sdist.merge_slab_sums(mine, x0 * 15, cnt * 15, piece_bytes=piece_bytes)
        covered = torch.zeros(n_rows, dtype=torch.int32)
        for f, c in stripes:
            assert torch.equal(mine["weight"][f:f + c], total["weight"][f:f + c])
            torch.testing.assert_close(mine["clip_feat"][f:f + c], total["clip_feat"][f:f + c], rtol=1e-6, atol=1e-6)
            covered[f:f + c] += 1
        allc = [torch.empty_like(covered) for _ in range(world)]
        dist.all_gather(allc, covered)
        assert torch.equal(torch.stack(allc).sum(0), torch.ones(n_rows, dtype=torch.int32)), "the ranks' stripes must tile the volume"
        out[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,piece_bytes", [(2, None), (4, 6 * 4 * 8), (8, 6 * 4 * 8)])
def test_slab_wise_merge_leaves_every_rank_its_part_of_every_slab(world, piece_bytes):
    """The slab-pipelined merge (distributed.fuse_merge_pipelined): each slab's rows are reduce-scattered on their own, piece
    by piece and in place; rank k must end with the exact sums of the k-th part of every piece of every slab, and the ranks'
    stripes must tile the volume."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_slab_merge_worker, args=(world, port, out, piece_bytes), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world))


def test_slab_bounds_tile_the_x_axis():
    for nx, s in ((256, 8), (127, 8), (48, 8), (5, 8), (64, 3)):
        b = sdist.slab_bounds(nx, s)
        assert b[0][0] == 0 and sum(c for _, c in b) == nx
        assert all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
        if nx % 16 == 0 and nx // 16 >= s:
            assert all(x % 16 == 0 and c % 16 == 0 for x, c in b)


def test_stripe_plan_tiles_and_ramp():
    """The layout arithmetic by itself: stripes of all ranks tile the rows exactly once for every world size and piece size;
    ramped slabs are small at both ends, aligned, and tile the x axis."""
    for n, world, piece in ((1000, 8, 64), (1000, 8, 8), (7, 8, 8), (4096, 4, 1024), (315, 2, 40), (16777216, 8, 2097152)):
        plan = sdist.stripe_plan(n, world, piece, row0=11)
        assert plan[0][0] == 11 and sum(r for _, r, _ in plan) == n
        seen = []
        for k in range(world):
            for f, c in sdist.stripes_of_rank(plan, k, world):
                assert c > 0
                seen.append((f, c))
        seen.sort()
        assert seen[0][0] == 11 and all(seen[i][0] + seen[i][1] == seen[i + 1][0] for i in range(len(seen) - 1))
        assert seen[-1][0] + seen[-1][1] == 11 + n
    assert sdist.piece_rows_for(2048, 8) == sdist._PIECE_BYTES // 2048 and sdist.piece_rows_for(1 << 40, 8) == 8
    b = sdist.slab_bounds(256, 8, ramp=True)
    assert [c for _, c in b] == [16, 32, 32, 48, 48, 32, 32, 16] and b[0][0] == 0 and sum(c for _, c in b) == 256
    for nx, k in ((256, 8), (128, 8), (64, 4), (48, 3), (33, 5), (16, 8), (255, 8)):
        b = sdist.slab_bounds(nx, k, ramp=True)
        assert b[0][0] == 0 and sum(c for _, c in b) == nx and all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
        assert all(c > 0 for _, c in b) and b[0][1] <= max(c for _, c in b) and b[-1][1] <= max(c for _, c in b)


def _gather_shards_worker(rank, world, port, out, with_plan=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 157

        class Vol:  # the attributes gather_shards reads
            pass

        v = Vol()
        full = {"clip_feat": torch.arange(n * 3, dtype=torch.float32).view(n, 3), "tsdf": torch.arange(n, dtype=torch.float32) * 0.5,
                "weight": torch.arange(n, dtype=torch.int32)}
        plan = sdist.stripe_plan(n, world, 8 * world)
        mine = sdist.stripes_of_rank(plan, rank, world)
        for k, t in full.items():
            x = torch.full_like(t, -1)  # rows a rank does not own hold garbage
            for f, c in mine:
                x[f:f + c] = t[f:f + c]
            setattr(v, k, x)
        v._shard_stripes = mine
        calls = {"all_gather_into_tensor": 0, "broadcast": 0}
        real = {k: getattr(dist, k) for k in calls}
        for k in calls:
            def counted(*a, _k=k, **kw):
                calls[_k] += 1
                return real[_k](*a, **kw)
            setattr(dist, k, counted)
        if with_plan:  # what merge_volumes / fuse_merge_pipelined record: one plan per merge (two here: as after two slabs)
            half = (len(plan) // 2)
            v._shard_plans = [plan[:half], plan[half:]]
        assert sdist.gather_shards(v) == [(0, n)] and v._shard_stripes is None
        for k in calls:
            setattr(dist, k, real[k])
        ok = all(torch.equal(getattr(v, k), t) for k, t in full.items())
        if with_plan:  # the striped in-place all-gather: one call per whole piece and tensor, a broadcast only for the tail
            whole_pieces = sum(1 for _, rows, c in plan if c > 0)
            ok = ok and calls["all_gather_into_tensor"] == 3 * whole_pieces and calls["broadcast"] <= 3 and v._shard_plans is None
        else:
            ok = ok and calls["all_gather_into_tensor"] == 0 and calls["broadcast"] > 0
        out[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,with_plan", [(2, False), (4, False), (2, True), (4, True), (8, True)])
def test_gather_shards_makes_a_striped_volume_whole(world, with_plan):
    """With the stripe plans the merge recorded: the in-place striped all-gather (every link busy); without (stripes set by
    hand): stripe lists exchanged, one rooted broadcast per stripe."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gather_shards_worker, args=(world, _free_port(), out, with_plan), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)
