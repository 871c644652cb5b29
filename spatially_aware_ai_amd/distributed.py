"""Multi-GPU fusion (new capability; the reference is single-device): frame-sharded with one merge (BASELINE config 4),
or voxel-sharded with no merge at all (``slab_of_rank`` / ``gather_frames``: every rank fuses EVERY frame into its own
x-slab of the volume -- a fusion module built with ``index_offset`` -- so the only exchange is the frames themselves and
the volume is born sharded the way ``query_sharded`` reads it).

Frame-sharded:

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm).  Frames
are independent units, so rank r fuses its contiguous block of frames into a private volume kept
as SUMS (``accum_mode = SAF_SUM``: F = sum of feature samples, C = sum of rgb, T = sum of clamped
sdf, integer weights / label counts).  A per-voxel running mean equals sum / count, so the merge
is one element-wise SUM across ranks followed by a local divide:

  * ``mode="reduce_scatter"`` (default): rank k receives the reduced voxel range k -- every xGMI
    link carries 1/world of the volume once, and the volume stays voxel-sharded, which is what the
    sharded text-query scan wants; ``gather=True`` adds the all-gather.
  * ``mode="all_reduce"``: every rank ends with the full merged volume.

Integer tensors merge exactly; the valid sets are per frame, hence unaffected by sharding.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _abi
from ._lib import SafError, check, current_stream_ptr, lib

_CHUNK_ELEMS = 1 << 28  # collectives are issued in <= 1 GiB (fp32) pieces


def shard_frames(n_frames: int, rank: int, world: int) -> range:
    """Contiguous block of frames owned by ``rank`` (first ranks take the remainder)."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def voxel_shard(n_voxels: int, rank: int, world: int):
    """(first, count) of the voxel range rank owns after a reduce-scatter merge: equal blocks of
    n_voxels // world, the remainder goes to the last rank."""
    per = n_voxels // world
    first = rank * per
    count = per if rank < world - 1 else n_voxels - first
    return first, count


def _chunks(t: torch.Tensor):
    flat = t.reshape(-1)
    for s in range(0, flat.numel(), _CHUNK_ELEMS):
        yield flat[s : s + _CHUNK_ELEMS]


def _all_reduce(t, group):
    for c in _chunks(t):
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)


def _rows_per_piece(t, world):
    """Rows of one rank block that one collective moves: world * rows * row_bytes <= _CHUNK_ELEMS * 4 bytes."""
    row_elems = max(1, t[0].numel()) if t.dim() > 1 else 1
    return max(1, _CHUNK_ELEMS // (world * row_elems))


def _reduce_scatter_rows(t, group, rank, world, _force_collective=False):
    """Sum dim-0 row blocks across ranks so that rank k holds the reduced rows of voxel_shard(k);
    other rows keep this rank's partial sums.

    nccl (RCCL): issued in pieces of at most 1 GiB.  A piece is rows [j, j + c) of EVERY rank's block; RCCL wants the send
    buffer contiguous, so the world slices are copied into a staging buffer (one strided device copy of 1 GiB, reused),
    ``reduce_scatter_tensor`` sums them straight into this rank's slice of the volume.  (The earlier form handed the whole
    34 GB volume to one in-place collective.)"""
    n = t.shape[0]
    per = n // world
    backend = dist.get_backend(group)
    if per > 0:
        main = t[: per * world]
        if backend == "nccl" or _force_collective:
            blocks = main.view((world, per) + tuple(main.shape[1:]))
            c = min(per, _rows_per_piece(t, world))
            stage = torch.empty((world, c) + tuple(main.shape[1:]), dtype=t.dtype, device=t.device)
            for j in range(0, per, c):
                cj = min(c, per - j)
                src = stage[:, :cj] if cj == c else stage.view(-1)[: world * cj * blocks[0, 0].numel()].view((world, cj) + tuple(main.shape[1:]))
                src.copy_(blocks[:, j : j + cj])
                dist.reduce_scatter_tensor(blocks[rank, j : j + cj], src, op=dist.ReduceOp.SUM, group=group)
        else:  # gloo has no reduce_scatter: one rooted reduce per destination
            for k in range(world):
                dist.reduce(main[k * per : (k + 1) * per], dst=dist.get_global_rank(group, k) if group else k,
                            op=dist.ReduceOp.SUM, group=group)
    if n > per * world:  # remainder rows belong to the last rank
        tail = t[per * world :]
        dst = world - 1
        dist.reduce(tail, dst=dist.get_global_rank(group, dst) if group else dst, op=dist.ReduceOp.SUM, group=group)


def _all_gather_rows(t, group, rank, world, _force_collective=False):
    """Every rank's reduced block to every rank, in pieces of at most 1 GiB (nccl: ``all_gather_into_tensor`` into a
    staging buffer, one strided copy back into the world blocks)."""
    n = t.shape[0]
    per = n // world
    if per > 0:
        main = t[: per * world]
        if dist.get_backend(group) == "nccl" or _force_collective:
            blocks = main.view((world, per) + tuple(main.shape[1:]))
            c = min(per, _rows_per_piece(t, world))
            stage = torch.empty((world, c) + tuple(main.shape[1:]), dtype=t.dtype, device=t.device)
            for j in range(0, per, c):
                cj = min(c, per - j)
                dst = stage[:, :cj] if cj == c else stage.view(-1)[: world * cj * blocks[0, 0].numel()].view((world, cj) + tuple(main.shape[1:]))
                dist.all_gather_into_tensor(dst, blocks[rank, j : j + cj].contiguous(), group=group)
                blocks[:, j : j + cj].copy_(dst)
        else:
            outs = [main[k * per : (k + 1) * per] for k in range(world)]
            dist.all_gather(outs, main[rank * per : (rank + 1) * per].clone(), group=group)
    if n > per * world:
        src = world - 1
        dist.broadcast(t[per * world :], src=dist.get_global_rank(group, src) if group else src, group=group)


def probe_collectives(device, group=None):
    """Run the reduce-scatter / all-gather forms ``merge_sums`` would use on a small tensor and check the sums: decides the
    merge mode BEFORE any volume is touched (a collective that raises half-way through a volume cannot be retried with
    another one -- the rows that were already summed would be summed twice).  Returns None, or what went wrong."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    try:
        n = 4 * world + 3
        t = (torch.arange(n * 6, dtype=torch.float32, device=device).view(n, 6) + 1.0) * (rank + 1)
        want = (torch.arange(n * 6, dtype=torch.float32, device=device).view(n, 6) + 1.0) * (world * (world + 1) / 2)
        _reduce_scatter_rows(t, group, rank, world)
        first, count = voxel_shard(n, rank, world)
        if not torch.equal(t[first : first + count], want[first : first + count]):
            return "reduce-scatter probe: wrong sums"
        _all_gather_rows(t, group, rank, world)
        if not torch.equal(t, want):
            return "all-gather probe: wrong rows"
    except Exception as e:  # noqa: BLE001 -- whatever the backend raises is the answer
        return f"{type(e).__name__}: {e}"[:200]
    return None


def slab_of_rank(nx: int, rank: int, world: int):
    """(first x index, number of x planes) of the volume slab rank owns in the VOXEL-sharded job: x-planes are whole
    contiguous ranges of the flat voxel index n = (x*ny + y)*nz + z, so slab k is voxel range k."""
    base, rem = divmod(nx, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def slab_planes_of_rank(nx: int, rank: int, world: int, block: int = 16):
    """The x-planes rank owns in the voxel-sharded job, BALANCED: cameras look at the middle of the scene, so a contiguous
    middle slab sees half again as many hits as an outer one (measured 111 vs 73 ms per job at 8 ranks).  The grid is cut
    into blocks of ``block`` x-planes (16: the tile of the row kernel's unit order and four bricks of the classification)
    and block b goes to rank (b mod B/2) mod world: a rank's block in the left half is as far out as its block in the right
    half is far in.  Falls back to the contiguous slab of ``slab_of_rank`` when the blocks do not divide evenly.  Returns
    the x indices, ascending."""
    if block % 4 != 0:
        raise ValueError("block must be a multiple of 4 (the classification's bricks are 4 x-planes wide)")
    nb = nx // block
    if nx % block != 0 or nb % (2 * world) != 0:
        first, cnt = slab_of_rank(nx, rank, world)
        return torch.arange(first, first + cnt)
    mine = [b for b in range(nb) if (b % (nb // 2)) % world == rank]
    return torch.cat([torch.arange(b * block, (b + 1) * block) for b in mine])


def gather_frames(tensors, group=None):
    """The exchange step of the voxel-sharded job: every rank contributes its frames (depth, rgb, poses, K, feature maps,
    [label maps]) and receives everybody's, in rank order -- 5 MB per 640x480 frame instead of the 34 GB of a volume merge.
    ``tensors``: a sequence of [F, ...] tensors (None entries stay None); ranks must hold the same number of frames."""
    world = dist.get_world_size(group)
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        t = t.contiguous()
        full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if dist.get_backend(group) == "nccl":
            out_blocks = full.view((world, t.shape[0]) + tuple(t.shape[1:]))
            c = min(t.shape[0], _rows_per_piece(t, world))
            if c == t.shape[0]:
                dist.all_gather_into_tensor(full, t, group=group)
            else:  # pieces of at most 1 GiB: a staging buffer per piece, one strided copy into the rank blocks
                stage = torch.empty((world, c) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                for j in range(0, t.shape[0], c):
                    cj = min(c, t.shape[0] - j)
                    dst = stage.view(-1)[: world * cj * t[0].numel()].view((world, cj) + tuple(t.shape[1:]))
                    dist.all_gather_into_tensor(dst, t[j : j + cj], group=group)
                    out_blocks[:, j : j + cj].copy_(dst)
        else:
            dist.all_gather(list(full.split(t.shape[0])), t, group=group)
        out.append(full)
    return out


VOLUME_TENSORS = ("clip_feat", "rgb", "tsdf", "weight", "tsdf_weight", "labels_one_hot")


# ---- the frame-sharded job with the merge PIPELINED slab by slab inside one job (BASELINE config 4's own layout) ----
#
# A serial job is fuse (84 ms) + one collective over the 34 GB volume (about 100 ms): 3.7 x at 8 GPUs.  The fused kernels are
# voxel-major and a slab of x-planes is a contiguous range of the flat voxel index, so the rank's frames can be fused SLAB
# BY SLAB into the one per-rank volume: when slab s is complete nothing will touch it again and its reduce-scatter can run on
# a communication stream while slab s + 1 is being fused.  Only the last slab's collective is exposed.  Rank k ends with
# the k-th part of EVERY slab (stripes), which is as good a voxel sharding for the query scan as one range.


def slab_bounds(nx: int, n_slabs: int, align: int = 16):
    """x-plane ranges [(x0, count)] of n_slabs slabs; boundaries on multiples of ``align`` planes (the row kernel's tile)
    when nx allows it."""
    unit = align if nx % align == 0 and nx // align >= n_slabs else 1
    blocks = nx // unit
    out, x = [], 0
    for s in range(n_slabs):
        cnt = (blocks // n_slabs + (1 if s < blocks % n_slabs else 0)) * unit
        if cnt:
            out.append((x, cnt))
        x += cnt
    return out


def slab_descriptor(fusion, x0: int, count: int):
    """The saf_volume descriptor of x-planes [x0, x0 + count) of ``fusion``'s volume: the same buffers, offset; fusing
    into it touches only that slab (the axis table starts at x0, so voxel centres -- and every decision -- are those of the
    full volume's voxels, bit for bit)."""
    vol = fusion._c_volume(for_fuse=True)
    nx, ny, nz = (int(v) for v in fusion.nvox)
    if not (0 <= x0 and count > 0 and x0 + count <= nx):
        raise ValueError("slab outside the volume")
    rows = x0 * ny * nz
    esz = 2 if fusion._buffers["clip_feat"].dtype == torch.bfloat16 else 4
    out = _abi.SafVolume.from_buffer_copy(vol)
    out.nx = count
    out.axis_x = vol.axis_x + 4 * x0
    out.tsdf = vol.tsdf + 4 * rows
    out.tsdf_weight = vol.tsdf_weight + 4 * rows
    out.weight = vol.weight + 4 * rows
    out.rgb = vol.rgb + 12 * rows
    out.clip_feat = vol.clip_feat + esz * int(fusion.n_clip_feats) * rows
    if vol.labels_one_hot:
        out.labels_one_hot = vol.labels_one_hot + 4 * int(vol.n_classes) * rows
    return out


def slab_rows(fusion, x0: int, count: int):
    nx, ny, nz = (int(v) for v in fusion.nvox)
    return x0 * ny * nz, count * ny * nz


def merge_slab_sums(tensors: dict, first_row: int, n_rows: int, group=None, mode: str = "reduce_scatter"):
    """SUM rows [first_row, first_row + n_rows) of every tensor across the ranks; with ``reduce_scatter`` rank k ends with
    the k-th part of the slab (returned as (first, count) in volume rows), the other rows keep partial sums."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return first_row, n_rows
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    for t in tensors.values():
        part = t[first_row : first_row + n_rows]
        if mode == "all_reduce":
            _all_reduce(part, group)
        else:
            _reduce_scatter_rows(part, group, rank, world)
    if mode == "all_reduce":
        return first_row, n_rows
    f, c = voxel_shard(n_rows, rank, world)
    return first_row + f, c


def fuse_merge_pipelined(fusion, frame_arr, n_frames, workspace, n_slabs=8, group=None, comm_stream=None, mode="reduce_scatter",
                         stats_ptr=None, profiler=None):
    """One frame-sharded job with the merge hidden behind the fusion: the rank's frames are fused slab by slab (SUM mode)
    and each finished slab is reduce-scattered + finalised on ``comm_stream`` while the next one is fused.  Returns the list
    of (first_row, count) stripes that hold final means on this rank.  The caller's stream ends ordered after the last
    collective."""
    if fusion.accum_mode != _abi.SAF_SUM:
        raise SafError("fuse_merge_pipelined fuses sums: set accum_mode = SAF_SUM (reset(accum_mode=SAF_SUM))")
    L = lib()
    dev = fusion._buffers["tsdf"].device
    main = torch.cuda.current_stream(dev)
    comm = comm_stream if comm_stream is not None else main
    tensors = {k: fusion._buffers[k] for k in VOLUME_TENSORS if fusion._buffers.get(k) is not None}
    stats_ptr = fusion._buffers["fuse_stats"].data_ptr() if stats_ptr is None else stats_ptr
    if fusion.__dict__.get("_feat_stale"):
        fusion.flush()  # (slabs are fused one by one: the deferred clear of reset() is finished first)
    stripes = []
    nx = int(fusion.nvox[0])
    for x0, cnt in slab_bounds(nx, n_slabs):
        vol = slab_descriptor(fusion, x0, cnt)
        check(L.saf_fuse_frames_profiled(C.byref(vol), frame_arr, n_frames, workspace.data_ptr(), workspace.numel(), stats_ptr,
                                         profiler, main.cuda_stream), "saf_fuse_frames (slab)")
        fused = main.record_event()
        with torch.cuda.stream(comm):
            comm.wait_event(fused)
            r0, nr = slab_rows(fusion, x0, cnt)
            first, count = merge_slab_sums(tensors, r0, nr, group, mode)
            vdesc = fusion._c_volume(for_fuse=True)
            check(L.saf_merge_finalize(C.byref(vdesc), first, count, comm.cuda_stream), "saf_merge_finalize")
            stripes.append((first, count))
    if comm is not main:
        main.wait_event(comm.record_event())
    return stripes


def merge_sums(tensors: dict, group=None, mode: str = "reduce_scatter", gather: bool = False):
    """Element-wise SUM of per-rank volume tensors (dict name -> tensor with voxels on dim 0).
    Device-agnostic (RCCL on GPUs, gloo in the CPU tests).  Returns (first, count): the voxel range
    that is fully reduced on this rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = next(iter(tensors.values())).shape[0]
    if world == 1:
        return 0, n
    if mode == "all_reduce":
        for t in tensors.values():
            _all_reduce(t, group)
        return 0, n
    if mode != "reduce_scatter":
        raise ValueError(mode)
    for t in tensors.values():
        _reduce_scatter_rows(t, group, rank, world)
    if gather:
        for t in tensors.values():
            _all_gather_rows(t, group, rank, world)
        return 0, n
    return voxel_shard(n, rank, world)


def _volume_tensors(fusion):
    return {k: getattr(fusion, k) for k in VOLUME_TENSORS if getattr(fusion, k, None) is not None}


def finalize_sums(fusion, first: int = 0, count: int | None = None):
    """sums -> means on the HIP device over voxels [first, first+count) (saf_merge_finalize); when
    the whole volume is covered the module returns to running-mean mode."""
    n = fusion.tsdf.numel()
    count = n - first if count is None else count
    vol = fusion._c_volume()
    with torch.cuda.device(fusion.tsdf.device):
        check(lib().saf_merge_finalize(C.byref(vol), first, count, current_stream_ptr()), "saf_merge_finalize")
    if first == 0 and count == n:
        fusion.accum_mode = _abi.SAF_RUNNING_MEAN


def means_to_sums(fusion):
    """Turn a running-mean volume into sums (x * w) so it can enter the reduction."""
    vol = fusion._c_volume()
    with torch.cuda.device(fusion.tsdf.device):
        check(lib().saf_mean_to_sum(C.byref(vol), 0, fusion.tsdf.numel(), current_stream_ptr()), "saf_mean_to_sum")
    fusion.accum_mode = _abi.SAF_SUM


def merge_volumes(fusion, group=None, mode: str = "reduce_scatter", gather: bool = False):
    """The single merge step of the frame-sharded job: RCCL SUM of the per-rank sum volumes, then
    the local divide.  ``fusion`` must have been fused with ``accum_mode = SAF_SUM`` (call
    ``means_to_sums`` first otherwise).  Returns the (first, count) voxel range that holds final
    means on this rank."""
    if fusion.clip_feat.dtype != torch.float32:
        raise SafError(
            f"merge_volumes needs an f32 feature volume (got {fusion.clip_feat.dtype}): per-rank SUMS kept in bf16 would "
            "round every addition to 8 bits; fuse the per-rank shards in f32 and convert after the merge"
        )
    if getattr(fusion, "_shard_range", None) is not None:
        raise SafError("this volume was already merged and holds only its voxel shard")
    flush = getattr(fusion, "flush", None)
    if flush is not None:
        flush()  # frames still queued behind integrate() belong to this job
    if fusion.accum_mode != _abi.SAF_SUM:
        means_to_sums(fusion)
    tensors = _volume_tensors(fusion)
    n = fusion.tsdf.numel()
    first, count = merge_sums(tensors, group=group, mode=mode, gather=False)
    finalize_sums(fusion, first, count)
    if (first, count) == (0, n):
        return first, count
    if gather:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        for t in tensors.values():
            _all_gather_rows(t, group, rank, world)
        fusion.accum_mode = _abi.SAF_RUNNING_MEAN
        return 0, n
    # voxel-sharded result: only [first, first+count) holds the job's means; the other rows hold this rank's
    # partial sums.  integrate() refuses to fuse into it (clipfusion._fuse) until gather_shards() is called.
    fusion.accum_mode = _abi.SAF_RUNNING_MEAN
    fusion._shard_range = (first, count)
    return first, count


def gather_shards(fusion, group=None):
    """All-gather a voxel-sharded merged volume (merge_volumes(..., gather=False)) so that every rank holds
    the whole merged volume and may fuse further frames."""
    if getattr(fusion, "_shard_range", None) is None:
        return 0, fusion.tsdf.numel()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    for t in _volume_tensors(fusion).values():
        _all_gather_rows(t, group, rank, world)
    fusion._shard_range = None
    return 0, fusion.tsdf.numel()


# --------------------------------------------------------------------------------------------
# voxel-sharded text-query scan (BASELINE config 5)
# --------------------------------------------------------------------------------------------


def shard_features_16(fusion, first, count, dtype=torch.float16):
    """The 16-bit copy of a volume's voxel shard that the wide scan reads (cached on the module until the volume
    is fused into or reset again).  One pass over count * D * 4 bytes."""
    key = (first, count, dtype, fusion.clip_feat.data_ptr(), int(fusion.fuse_stats[2]))
    cached = fusion.__dict__.get("_shard16")
    if cached is not None and cached[0] == key:
        return cached[1]
    feats = fusion.clip_feat[first:first + count]
    if feats.dtype != dtype:
        out = torch.empty(feats.shape, dtype=dtype, device=feats.device)
        step = max(1, (1 << 26) // max(1, feats.shape[1]))  # in pieces: no second full-size temporary
        for s0 in range(0, feats.shape[0], step):
            out[s0:s0 + step].copy_(feats[s0:s0 + step])
        feats = out
    fusion.__dict__["_shard16"] = (key, feats)
    return feats


def query_sharded(fusion, text, epilogue="row_argmax", group=None, gather=True, dtype=torch.float16, scan_fn=None, **kw):
    """BASELINE config 5: many text queries over the merged volume, voxel-sharded over the ranks.

    After ``merge_volumes(..., gather=False)`` rank k holds the means of voxel range k (its ``_shard_range``); a
    volume that is whole on every rank is split by ``voxel_shard``.  Each rank scans ONLY its range with
    ``query_scan_wide`` (no data-path collective: the volume never moves) and the small results are combined:

    * ``"query_max"``   -> (value [Q], voxel [Q]) identical on every rank: an all-gather of Q (score, voxel) pairs and a
      max -- equal scores resolve to the smaller voxel index, as a single-rank scan would;
    * ``"row_argmax"``  -> (index, value) of this rank's voxel range, or of all N voxels when ``gather`` (all-gather of
      4 + 4 bytes per voxel);
    * ``"scores"`` / ``"vs_background"`` -> this rank's [count, Q] block (33 GB in all at config 5: it stays sharded).

    ``scan_fn(feats, text, epilogue, row_offset=..., **kw)`` defaults to the HIP scan; the CPU tests inject an
    oracle-backed one to exercise the sharding and the reductions under gloo."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = fusion.tsdf.numel()
    rng = getattr(fusion, "_shard_range", None)
    first, count = rng if rng is not None else voxel_shard(n, rank, world)
    if scan_fn is None:
        from .clipfusion import query_scan_wide

        scan_fn = query_scan_wide
        feats = shard_features_16(fusion, first, count, dtype)
    else:
        feats = fusion.clip_feat[first:first + count]
    res = scan_fn(feats, text, epilogue, row_offset=first, **kw)
    if world == 1:
        return res
    if epilogue == "query_max":
        val, row = res
        vals = [torch.empty_like(val) for _ in range(world)]
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(vals, val.contiguous(), group=group)
        dist.all_gather(rows, row.contiguous(), group=group)
        vals, rows = torch.stack(vals), torch.stack(rows)
        best = vals.max(dim=0).values
        # among the ranks that reach the maximum, the smallest voxel index (ranks own ascending ranges)
        big = torch.iinfo(torch.int64).max
        cand = torch.where((vals == best[None]) & (rows >= 0), rows, torch.full_like(rows, big))
        pick = cand.min(dim=0).values
        return best, torch.where(pick == big, torch.full_like(pick, -1), pick)
    if epilogue == "row_argmax" and gather:
        idx, val = res
        counts = [voxel_shard(n, k, world)[1] for k in range(world)] if rng is None else None
        if counts is None:  # ranges came from the merge: exchange their sizes
            c = torch.tensor([count], dtype=torch.int64, device=idx.device)
            cs = [torch.empty_like(c) for _ in range(world)]
            dist.all_gather(cs, c, group=group)
            counts = [int(x) for x in cs]
        pad = max(counts)
        out = []
        for t in (idx, val):
            mine = torch.zeros(pad, dtype=t.dtype, device=t.device)
            mine[:count] = t
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            out.append(torch.cat([p[:c] for p, c in zip(parts, counts)]))
        return tuple(out)
    return res
