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

_CHUNK_ELEMS = 1 << 28  # all_reduce / frame exchange are issued in <= 1 GiB (fp32) pieces
_PIECE_BYTES = 1 << 30  # the striped reduce-scatter / all-gather work IN PLACE (no staging copy): pieces of <= 1 GiB (byte counts stay below 2^31 whatever the collective library does with them)


def shard_frames(n_frames: int, rank: int, world: int) -> range:
    """Contiguous block of frames owned by ``rank`` (first ranks take the remainder)."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def voxel_shard(n_voxels: int, rank: int, world: int):
    """(first, count) of rank's part when n_voxels are cut into ``world`` equal contiguous blocks (the remainder goes to the
    last rank): how ``query_sharded`` splits a volume that is whole on every rank, and the ownership inside ONE piece of
    the striped merge (``stripe_plan``)."""
    per = n_voxels // world
    first = rank * per
    count = per if rank < world - 1 else n_voxels - first
    return first, count


# ---- striped ownership: the merge's collectives run in place, piece by piece ---------------------------------------------
#
# RCCL's reduce-scatter wants ONE contiguous send buffer of world equal parts and leaves part k on rank k.  Round 3 kept
# "rank k owns voxel range k of the whole volume", which made every <= 1 GiB piece a strided gather of world slices into a
# staging buffer (+ 34 GB read + 34 GB written per merge, and one staging buffer serialising copy j + 1 behind collective j).
# Now the OWNERSHIP follows the pieces: the rows of a volume (or of one slab of it) are cut into contiguous pieces of
# `piece_rows` rows (a multiple of world; <= 1 GiB of the widest tensor), and inside every piece rank k owns the k-th of
# world equal parts.  A piece is then exactly what reduce_scatter_tensor / all_gather_into_tensor take in place:
#   reduce_scatter_tensor(piece[k c : (k + 1) c], piece)        all_gather_into_tensor(piece, piece[k c : (k + 1) c])
# No staging buffer, no copy.  Rank k ends with a list of STRIPES, one per piece; every tensor of the volume uses the same
# row boundaries, so a voxel's features, colours, weights and label counts are finalised on the same rank.


def piece_rows_for(row_bytes: int, world: int, piece_bytes: int | None = None) -> int:
    """Rows of one piece for tensors whose widest row has ``row_bytes`` bytes: a multiple of world, at least world."""
    piece_bytes = _PIECE_BYTES if piece_bytes is None else piece_bytes
    return max(world, (piece_bytes // max(1, row_bytes)) // world * world)


def stripe_plan(n_rows: int, world: int, piece_rows: int, row0: int = 0):
    """[(first_row, rows, c)] of the pieces that tile rows [row0, row0 + n_rows): inside a piece rank k owns rows
    [first_row + k c, first_row + (k + 1) c); the piece's last rows - world c (fewer than world, and only in the LAST piece)
    belong to the last rank."""
    if piece_rows % world != 0 or piece_rows <= 0:
        raise ValueError("piece_rows must be a positive multiple of the world size")
    plan, r = [], 0
    while r < n_rows:
        rows = min(piece_rows, n_rows - r)
        plan.append((row0 + r, rows, rows // world))
        r += rows
    return plan


def stripes_of_rank(plan, rank: int, world: int):
    """The (first, count) row ranges ``rank`` owns under ``plan`` (adjacent ranges merged), ascending."""
    out = []

    def add(f, c):
        if c <= 0:
            return
        if out and out[-1][0] + out[-1][1] == f:
            out[-1] = (out[-1][0], out[-1][1] + c)
        else:
            out.append((f, c))

    for first, rows, c in plan:
        add(first + rank * c, c)
        if rank == world - 1:
            add(first + world * c, rows - world * c)
    return out


def _chunks(t: torch.Tensor):
    flat = t.reshape(-1)
    for s in range(0, flat.numel(), _CHUNK_ELEMS):
        yield flat[s : s + _CHUNK_ELEMS]


def _all_reduce(t, group):
    for c in _chunks(t):
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)


def _rows_per_piece(t, world):
    """Rows of one rank block that one staged collective of the frame exchange moves (<= 1 GiB in all)."""
    row_elems = max(1, t[0].numel()) if t.dim() > 1 and t.shape[0] > 0 else 1
    return max(1, _CHUNK_ELEMS // (world * row_elems))


def _global_rank(group, k):
    return dist.get_global_rank(group, k) if group else k


def _reduce_scatter_striped(t, plan, group, rank, world):
    """SUM the rows of ``t`` across the ranks, piece by piece and in place: rank k ends with the reduced rows of its stripes
    (``stripes_of_rank``), the other rows keep this rank's partial sums.  The same calls on every backend (RCCL, gloo)."""
    for first, rows, c in plan:
        if c > 0:
            piece = t[first : first + world * c]
            dist.reduce_scatter_tensor(piece[rank * c : (rank + 1) * c], piece, op=dist.ReduceOp.SUM, group=group)
        if rows > world * c:  # fewer than world rows at the very end: they belong to the last rank
            dist.reduce(t[first + world * c : first + rows], dst=_global_rank(group, world - 1), op=dist.ReduceOp.SUM, group=group)


def _all_gather_striped(t, plan, group, rank, world):
    """Every rank's stripes to every rank, piece by piece and in place."""
    for first, rows, c in plan:
        if c > 0:
            piece = t[first : first + world * c]
            dist.all_gather_into_tensor(piece, piece[rank * c : (rank + 1) * c], group=group)
        if rows > world * c:
            dist.broadcast(t[first + world * c : first + rows], src=_global_rank(group, world - 1), group=group)


# ---- the SPARSE merge: rows no rank touched do not travel ---------------------------------------------------------------
#
# A feature row is non-zero only if its weight is (rows are written together with a weight increment, clipfusion.py:715-721),
# and a real scan touches a thin shell of the grid (the coherent scene: 17 % of the rows per window, a fifth of the volume
# over a whole job) -- while the merge of config 4 is collective-bound.  So `weight` -- 4 bytes per voxel,
# 0.2 % of the volume -- is all-reduced FIRST: every rank then knows the job's total weight, i.e. the union of the rows any
# rank touched, without exchanging a bitmap.  Per piece of the stripe plan the wide tensors (features, colours, label
# counts) then take one of two routes, decided identically on every rank from that mask:
#   * most rows touched (incoherent depth; > `sparse` of the piece): the in-place striped reduce-scatter, as before;
#   * few rows touched: the touched rows of part k are packed and sent to rank k (all_to_all_single with uneven splits -- on
#     the fully connected xGMI mesh every pair has its own link), rank k adds the world contributions in rank order and
#     writes the sums back to its rows.  Same ownership, same stripes, same result up to the order of the fp32 additions.
# tsdf / tsdf_weight are touched almost everywhere (every voxel in front of a surface) and are 8 bytes per voxel: always dense.
_SPARSE_TENSORS = ("clip_feat", "rgb", "labels_one_hot")
_SPARSE_DEFAULT = 0.5
last_merge = {"pieces": 0, "packed": 0, "rows": 0, "touched_rows": 0}  # of this process's latest _merge_rows (bench / tests)
_a2a_probe = {}  # (group, device type) -> None (the packed route's collective works) or what went wrong: probed ONCE per process


def sparse_threshold(sparse=None):
    """THIS rank's wish for the touched fraction of a piece up to which it travels packed: ``sparse`` if given, else
    SAF_MERGE_SPARSE (0 = the dense route always), else 0.5.  What a merge uses is ``agree_sparse``'s answer."""
    import os

    if sparse is None:
        e = os.environ.get("SAF_MERGE_SPARSE")
        sparse = float(e) if e not in (None, "") else _SPARSE_DEFAULT
    return max(0.0, float(sparse))


def sparse_wish(sparse, device, group=None):
    """THIS rank's threshold for the packed route, safe to offer: ``sparse_threshold`` -- but with ``sparse=None`` (the library
    default) only if ``probe_all_to_all`` succeeded, run once per process and group and cached: a collective that raises
    part-way through a volume cannot be retried (``probe_collectives``), so the route is tried on a small tensor before any
    volume is touched.  An explicit ``sparse`` means the caller has probed (``bench.py`` does) or knows its backend.  What a
    merge USES is the minimum of the ranks' wishes (``_merge_rows``): ranks whose SAF_MERGE_SPARSE differ would otherwise pick
    different collectives for the same piece and hang."""
    thr = sparse_threshold(sparse)
    if sparse is None:  # (every rank that was not told probes, whatever its own wish: the probe is itself a collective)
        key = (id(group) if group is not None else None, torch.device(device).type)
        if key not in _a2a_probe:
            _a2a_probe[key] = probe_all_to_all(device, group)
        if _a2a_probe[key] is not None:
            thr = 0.0
    return thr


class _Touched:
    """What one merge knows about the rows any rank touched (from the all-reduced weight of the rows the plan covers):
    ``offs[j][k]`` = position, among the touched rows in ascending order, of the first touched row of part k of piece j
    (k = world: the end of the piece's world equal parts; a tail of fewer than world rows is reduced on its own), after ONE
    host synchronisation.  Device tensors: a scan kernel and the boundaries' positions copied to pinned memory
    (``saf_merge_scan_touched``; no index list -- pack / add read the positions).  CPU tensors (the gloo tests): torch."""

    def __init__(self, weight_total, plan, world):
        self.first0 = plan[0][0]
        self.n_rows = plan[-1][0] + plan[-1][1] - self.first0
        self.w = weight_total[self.first0 : self.first0 + self.n_rows]
        self.world = world
        bounds = []
        for first, rows, c in plan:
            bounds += [first - self.first0 + k * c for k in range(world + 1)]
        self.hip = self.w.is_cuda
        self._idx = None
        if self.hip:
            L = lib()
            dev = self.w.device
            self.pos = torch.empty(self.n_rows + 1, dtype=torch.int32, device=dev)
            b = torch.tensor(bounds, dtype=torch.int64).to(dev, non_blocking=False)
            need = L.saf_merge_scan_workspace_bytes(self.n_rows, len(bounds))
            ws = torch.empty(max(1, need), dtype=torch.uint8, device=dev)
            host = torch.empty(len(bounds), dtype=torch.int32).pin_memory()
            with torch.cuda.device(dev):
                check(L.saf_merge_scan_touched(self.w.data_ptr(), self.n_rows, self.pos.data_ptr(), b.data_ptr(), len(bounds),
                                               host.data_ptr(), ws.data_ptr(), ws.numel(), current_stream_ptr()), "saf_merge_scan_touched")
                torch.cuda.current_stream(dev).synchronize()  # the one host sync of a merge: the collective's split sizes
            pre = host.tolist()
        else:
            touched = self.w > 0
            cs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(touched, 0, dtype=torch.int64)])
            pre = cs[torch.tensor(bounds, dtype=torch.int64)].tolist()
            self._touched = touched
        self.offs = [pre[j * (world + 1) : (j + 1) * (world + 1)] for j in range(len(plan))]

    def idx(self):  # (CPU route only: the touched rows' indices, relative to first0)
        if self._idx is None:
            self._idx = torch.nonzero(self._touched).squeeze(1)
        return self._idx


def _reduce_scatter_packed(t, touched, j, first, c, group, rank, world):
    """One piece, packed: the touched rows of part k go to rank k, which adds the world contributions (rank order:
    deterministic) and writes them back in place.  On the device the send buffer is packed and the received contributions
    are added by two HIP passes (``saf_merge_pack_rows`` / ``saf_merge_add_packed``); CPU tensors take the same steps in torch."""
    offs_j = touched.offs[j]
    counts = [offs_j[k + 1] - offs_j[k] for k in range(world)]
    total, mine = offs_j[world] - offs_j[0], counts[rank]
    if total == 0:
        return
    row_shape = tuple(t.shape[1:])
    view = t[touched.first0 : touched.first0 + touched.n_rows]
    rel = first - touched.first0
    send = torch.empty((total,) + row_shape, dtype=t.dtype, device=t.device)
    recv = torch.empty((world * mine,) + row_shape, dtype=t.dtype, device=t.device)
    if touched.hip:
        L = lib()
        row_bytes = (view[0].numel() if view.dim() > 1 else 1) * t.element_size()
        if not view.is_contiguous() or t.element_size() != 4:
            raise SafError("the packed merge moves contiguous rows of 4-byte elements")
        with torch.cuda.device(t.device):
            check(L.saf_merge_pack_rows(view.data_ptr(), row_bytes, touched.w.data_ptr(), touched.pos.data_ptr(), rel, world * c,
                                        send.data_ptr(), current_stream_ptr()), "saf_merge_pack_rows")
            dist.all_to_all_single(recv, send, [mine] * world, counts, group=group)
            check(L.saf_merge_add_packed(view.data_ptr(), row_bytes, 1 if t.dtype.is_floating_point else 0, touched.w.data_ptr(),
                                         touched.pos.data_ptr(), rel + rank * c, c, recv.data_ptr(), mine, world, current_stream_ptr()),
                  "saf_merge_add_packed")
        return
    idx = touched.idx()
    torch.index_select(view, 0, idx[offs_j[0] : offs_j[world]], out=send)
    dist.all_to_all_single(recv, send, [mine] * world, counts, group=group)
    if mine:
        parts = recv.view((world, mine) + row_shape)
        red = parts[0].clone()
        for k in range(1, world):  # rank order, as the device pass adds them
            red += parts[k]
        view.index_copy_(0, idx[offs_j[rank] : offs_j[rank + 1]], red)


def _merge_rows(tensors, plan, group, rank, world, wish):
    """SUM the rows the plan covers across the ranks; rank k ends with the reduced rows of its stripes, the other rows keep
    this rank's partial sums -- except ``weight``, which is ALL-REDUCED (whatever the route: 0.2 % of the volume): every rank
    ends with the job's total weight on every row, i.e. the union of the touched rows.  ``wish``: this rank's threshold for the
    packed route (``sparse_wish``); the ranks use the MINIMUM of their wishes (one 8-byte all_reduce, read in the same host
    synchronisation as the split sizes); pieces whose touched share is at most that travel packed.  Returns their number."""
    w = tensors.get("weight")
    wide = [k for k in _SPARSE_TENSORS if tensors.get(k) is not None]
    packed = 0
    last_merge.update(pieces=len(plan), packed=0, rows=sum(r for _, r, _ in plan), touched_rows=-1)
    if not plan:
        return 0
    if w is not None:
        first0 = plan[0][0]
        n_rows = plan[-1][0] + plan[-1][1] - first0
        _all_reduce(w[first0 : first0 + n_rows], group)
    rest = {k: t for k, t in tensors.items() if k != "weight"}
    thr = 0.0
    if w is not None and wide:
        agreed = torch.tensor([float(wish)], dtype=torch.float64, device=w.device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=group)
        if wish > 0:  # (a rank that wishes 0 knows the minimum without looking: no scan, no host sync)
            touched = _Touched(w, plan, world)
            thr = float(agreed)
    if thr > 0:
        for j, (first, rows, c) in enumerate(plan):
            frac = (touched.offs[j][world] - touched.offs[j][0]) / max(1, world * c)
            go_packed = c > 0 and frac <= thr
            packed += 1 if go_packed else 0
            for name, t in rest.items():
                if go_packed and name in wide:
                    _reduce_scatter_packed(t, touched, j, first, c, group, rank, world)
                    if rows > world * c:
                        dist.reduce(t[first + world * c : first + rows], dst=_global_rank(group, world - 1), op=dist.ReduceOp.SUM, group=group)
                else:
                    _reduce_scatter_striped(t, [(first, rows, c)], group, rank, world)
        last_merge.update(packed=packed, touched_rows=int(sum(o[world] - o[0] for o in touched.offs)))
        return packed
    for t in rest.values():
        _reduce_scatter_striped(t, plan, group, rank, world)
    return 0


def probe_all_to_all(device, group=None):
    """The packed route's collective on a small tensor with uneven splits, checked: None, or what went wrong (the ranks agree
    on the outcome).  A merge whose probe fails runs dense (``sparse=0``)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [1 + (k % 3) for k in range(world)]
    why = None
    try:
        send = torch.cat([torch.full((counts[k], 4), float(rank * 100 + k), device=device) for k in range(world)])
        mine = counts[rank]
        recv = torch.empty((world * mine, 4), device=device)
        dist.all_to_all_single(recv, send, [mine] * world, counts, group=group)
        want = torch.cat([torch.full((mine, 4), float(r * 100 + rank), device=device) for r in range(world)])
        if not torch.equal(recv, want):
            why = "all_to_all probe: wrong rows"
    except Exception as e:  # noqa: BLE001
        why = f"{type(e).__name__}: {e}"[:200]
    try:
        if not _agree(why is None, device, group):
            return why or "all_to_all probe failed on another rank"
    except Exception as e:  # noqa: BLE001
        return why or f"{type(e).__name__}: {e}"[:200]
    return None


def _plan_for(tensors: dict, n_rows: int, world: int, row0: int = 0, piece_bytes: int | None = None):
    widest = max(max(1, t[0].numel()) * t.element_size() if t.shape[0] else 1 for t in tensors.values())
    return stripe_plan(n_rows, world, piece_rows_for(widest, world, piece_bytes), row0)


def _agree(ok: bool, device, group) -> bool:
    """True when ``ok`` on EVERY rank (one small all_reduce: the ranks decide together before the next collective)."""
    flag = torch.tensor([0.0 if ok else 1.0], device=device)
    dist.all_reduce(flag, group=group)
    return float(flag) == 0.0


def probe_collectives(device, group=None):
    """Run the striped reduce-scatter / all-gather ``merge_sums`` would use on a small tensor -- several pieces, a ragged
    last piece, a tail of fewer than world rows -- and check the sums: decides the merge mode BEFORE any volume is touched
    (a collective that raises half-way through a volume cannot be retried with another one: the rows that were already
    summed would be summed twice).  After every step the ranks agree on the outcome with an all_reduce of a flag, so a rank
    that failed and a rank that did not never issue different collectives next.  Returns None, or what went wrong."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = 5 * world + world // 2 + 3  # three pieces of 2 * world rows; the last one ragged, with a tail
    base = torch.arange(n * 6, dtype=torch.float32, device=device).view(n, 6) + 1.0
    t, want = base * (rank + 1), base * (world * (world + 1) / 2)
    plan = stripe_plan(n, world, 2 * world)
    why = None
    try:
        _reduce_scatter_striped(t, plan, group, rank, world)
        if any(not torch.equal(t[f : f + c], want[f : f + c]) for f, c in stripes_of_rank(plan, rank, world)):
            why = "reduce-scatter probe: wrong sums"
    except Exception as e:  # noqa: BLE001 -- whatever the backend raises is the answer
        why = f"{type(e).__name__}: {e}"[:200]
    try:
        if not _agree(why is None, device, group):
            return why or "reduce-scatter probe failed on another rank"
    except Exception as e:  # noqa: BLE001
        return why or f"{type(e).__name__}: {e}"[:200]
    try:
        _all_gather_striped(t, plan, group, rank, world)
        if not torch.equal(t, want):
            why = "all-gather probe: wrong rows"
    except Exception as e:  # noqa: BLE001
        why = f"{type(e).__name__}: {e}"[:200]
    try:
        if not _agree(why is None, device, group):
            return why or "all-gather probe failed on another rank"
    except Exception as e:  # noqa: BLE001
        return why or f"{type(e).__name__}: {e}"[:200]
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
        if t.shape[0] == 0:  # nothing to exchange (every rank holds the same number of frames: none)
            out.append(full)
            continue
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


def slab_bounds(nx: int, n_slabs: int, align: int = 16, ramp: bool = False):
    """x-plane ranges [(x0, count)] of n_slabs slabs; boundaries on multiples of ``align`` planes (the row kernel's tile)
    when nx allows it.  ``ramp``: small slabs at both ends, large ones in the middle (256 planes, 8 slabs: 16 32 32 48 48 32
    32 16) -- the communication stream of the slab-pipelined merge starts after one SMALL slab's fusion instead of 1/8 of
    the job, and the collective left exposed at the end is a small slab's."""
    unit = align if nx % align == 0 and nx // align >= n_slabs else 1
    blocks = nx // unit
    n_slabs = max(1, min(n_slabs, blocks)) if blocks else 0
    if ramp and n_slabs >= 3 and blocks > n_slabs:
        w = [min(i + 1, n_slabs - i) for i in range(n_slabs)]
        ideal = [blocks * x / sum(w) for x in w]
        cnts = [max(1, int(v + 0.5)) for v in ideal]
        mid_out = sorted(range(n_slabs), key=lambda i: abs(i - (n_slabs - 1) / 2))  # what rounding left over: middle slabs first
        k = 0
        while sum(cnts) < blocks:
            cnts[mid_out[k % n_slabs]] += 1
            k += 1
        k = 0
        while sum(cnts) > blocks:
            i = mid_out[k % n_slabs]
            if cnts[i] > 1:
                cnts[i] -= 1
            k += 1
    else:
        cnts = [blocks // n_slabs + (1 if i < blocks % n_slabs else 0) for i in range(n_slabs)] if n_slabs else []
    out, x = [], 0
    for c in cnts:
        if c:
            out.append((x, c * unit))
        x += c * unit
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


def merge_slab_sums(tensors: dict, first_row: int, n_rows: int, group=None, mode: str = "reduce_scatter", piece_bytes=None,
                    sparse=None, plans_out=None, wish=None):
    """SUM rows [first_row, first_row + n_rows) of every tensor across the ranks; with ``reduce_scatter`` rank k ends with
    its stripes of the slab (returned as a list of (first, count) in volume rows: the k-th part of every piece of the slab),
    the other rows keep partial sums (``weight``: the job's total on every row, whatever the route -- ``_merge_rows``).
    ``plans_out``: a list the slab's stripe plan is appended to (``gather_shards`` all-gathers along it).  ``wish``: this rank's
    threshold for the packed route when the caller has settled it for all slabs at once (``sparse_wish``); else it is settled here."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [(first_row, n_rows)]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if mode == "all_reduce":
        for t in tensors.values():
            _all_reduce(t[first_row : first_row + n_rows], group)
        return [(first_row, n_rows)]
    plan = _plan_for(tensors, n_rows, world, first_row, piece_bytes)
    wish = sparse_wish(sparse, next(iter(tensors.values())).device, group) if wish is None else wish
    _merge_rows(tensors, plan, group, rank, world, wish)
    if plans_out is not None:
        plans_out.append(plan)
    return stripes_of_rank(plan, rank, world)


def _require_f32_sums(fusion, what):
    if fusion._buffers["clip_feat"].dtype != torch.float32:
        raise SafError(
            f"{what} needs an f32 feature volume (got {fusion._buffers['clip_feat'].dtype}): per-rank SUMS kept in bf16 would "
            "round every addition to 8 bits; fuse the per-rank shards in f32 and convert after the merge"
        )


def fuse_merge_pipelined(fusion, frame_arr, n_frames, workspace, n_slabs=8, group=None, comm_stream=None, mode="reduce_scatter",
                         stats_ptr=None, profiler=None, ramp=True, sparse=None):
    """One frame-sharded job with the merge hidden behind the fusion: the rank's frames are fused slab by slab (SUM mode)
    and each finished slab is reduce-scattered (in place, striped: ``merge_slab_sums``) + finalised on ``comm_stream`` while
    the next one is fused.  Returns the list of (first_row, count) stripes that hold final means on this rank, and leaves the
    module in running-mean mode with those stripes recorded (``_shard_stripes``): it refuses further ``integrate`` calls
    until ``gather_shards``.  The caller's stream ends ordered after the last collective."""
    _require_f32_sums(fusion, "fuse_merge_pipelined")
    check_poisoned = getattr(fusion, "_check_poisoned", None)
    if check_poisoned is not None:
        check_poisoned()  # (a volume a failed flush or merge left half-done stays unusable until reset())
    if getattr(fusion, "_shard_stripes", None) is not None:
        raise SafError("this volume was already merged and holds only its voxel stripes")
    # frames still queued behind integrate() belong to the volume.  A volume fresh from a lazy reset() keeps its deferred clear:
    # the slab-wise call zeroes a slab's still-unwritten rows behind the slab's last row kernel (0.4 ms in all on the benchmark's
    # depth; the clear up front writes all 34 GB)
    recycled = bool(fusion.__dict__.get("_feat_stale")) and not fusion._queue_busy()
    if not recycled:
        fusion.flush()
    if fusion.accum_mode != _abi.SAF_SUM:
        raise SafError("fuse_merge_pipelined fuses sums: set accum_mode = SAF_SUM (reset(accum_mode=SAF_SUM))")
    L = lib()
    dev = fusion._buffers["tsdf"].device
    main = torch.cuda.current_stream(dev)
    comm = comm_stream if comm_stream is not None else main
    tensors = {k: fusion._buffers[k] for k in VOLUME_TENSORS if fusion._buffers.get(k) is not None}
    stats_ptr = fusion._buffers["fuse_stats"].data_ptr() if stats_ptr is None else stats_ptr
    stripes, plans = [], []
    nx = int(fusion.nvox[0])
    n = fusion._buffers["tsdf"].numel()
    slabs = slab_bounds(nx, n_slabs, ramp=ramp)
    k = len(slabs)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    # the packed route: probed (once per process) BEFORE anything touches the volume; one wish for every slab of this job
    wish = sparse_wish(sparse, dev, group) if world > 1 and mode == "reduce_scatter" else 0.0
    touched_volume = False  # has any kernel or collective of this job written to the volume?
    # ONE C call fuses every frame into slab 0, slab 1, ... (the first window of a slab is classified beside the last row
    # kernel of the slab before it) and records an event behind each slab; the collectives wait for those on `comm`
    try:
        with torch.cuda.device(dev):
            events = [torch.cuda.Event() for _ in slabs]
            for ev in events:
                ev.record(main)  # (torch creates the HIP event at its first record; the C call records it again, for real)
            x0s = (C.c_int32 * k)(*[s_[0] for s_ in slabs])
            nxs = (C.c_int32 * k)(*[s_[1] for s_ in slabs])
            handles = (C.c_void_p * k)(*[int(ev.cuda_event) for ev in events])
            vol = fusion._c_volume(for_fuse=True)
            rc = L.saf_fuse_frames_slabs(C.byref(vol), frame_arr, n_frames, x0s, nxs, k, handles, 1 if recycled else 0,
                                         workspace.data_ptr(), workspace.numel(), stats_ptr, profiler, main.cuda_stream)
            # (SAF_E_INVALID / SAF_E_WORKSPACE are rejections at the entry, before any kernel ran: the volume is as it was)
            touched_volume = rc == 0 or rc == _abi.SAF_E_HIP
            check(rc, "saf_fuse_frames_slabs")
            if recycled:  # (the slabs cover the volume: slab_bounds)
                fusion.__dict__["_feat_stale"] = False
            with torch.cuda.stream(comm):
                for (x0, cnt), ev in zip(slabs, events):
                    comm.wait_event(ev)
                    r0, nr = slab_rows(fusion, x0, cnt)
                    for first, count in merge_slab_sums(tensors, r0, nr, group, mode, plans_out=plans, wish=wish):
                        check(L.saf_merge_finalize(C.byref(vol), first, count, comm.cuda_stream), "saf_merge_finalize")
                        stripes.append((first, count))
    except BaseException:
        # some slabs are reduced / finalised, others not: the volume is neither sums nor means.  Poisoned until reset(),
        # like a failed flush (clipfusion._flush_pending): a later integrate or merge must not count anything twice.
        # (Not when the C call refused its arguments at the entry: nothing ran, the caller may fix them and call again.)
        if touched_volume:
            fusion.__dict__["_poisoned"] = "fuse_merge_pipelined failed part-way: the volume is half sums, half means"
        raise
    if comm is not main:
        main.wait_event(comm.record_event())
    # the volume now holds means on this rank's stripes (partial sums elsewhere): no longer a SUM volume
    torch.nn.Module.__setattr__(fusion, "accum_mode", _abi.SAF_RUNNING_MEAN)
    whole = sum(c for _, c in stripes) == n  # one rank, or all_reduce: every slab is complete here
    fusion._shard_stripes = None if whole else list(stripes)
    fusion.__dict__["_shard_plans"] = None if whole else plans
    return stripes


def merge_sums(tensors: dict, group=None, mode: str = "reduce_scatter", gather: bool = False, piece_bytes=None, sparse=None,
               plans_out=None):
    """Element-wise SUM of per-rank volume tensors (dict name -> tensor with voxels on dim 0).
    Device-agnostic (RCCL on GPUs, gloo in the CPU tests).  Returns the list of (first, count) voxel ranges that are fully
    reduced on this rank: its stripes (``stripe_plan``), or [(0, n)]."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = next(iter(tensors.values())).shape[0]
    if world == 1:
        return [(0, n)]
    if mode == "all_reduce":
        for t in tensors.values():
            _all_reduce(t, group)
        return [(0, n)]
    if mode != "reduce_scatter":
        raise ValueError(mode)
    plan = _plan_for(tensors, n, world, 0, piece_bytes)
    _merge_rows(tensors, plan, group, rank, world, sparse_wish(sparse, next(iter(tensors.values())).device, group))
    if gather:
        for t in tensors.values():
            _all_gather_striped(t, plan, group, rank, world)
        return [(0, n)]
    if plans_out is not None:
        plans_out.append(plan)
    return stripes_of_rank(plan, rank, world)


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


def merge_volumes(fusion, group=None, mode: str = "reduce_scatter", gather: bool = False, piece_bytes=None, sparse=None):
    """The single merge step of the frame-sharded job: RCCL SUM of the per-rank sum volumes, then
    the local divide.  ``fusion`` must have been fused with ``accum_mode = SAF_SUM`` (call
    ``means_to_sums`` first otherwise).  Returns the list of (first, count) voxel ranges that hold final
    means on this rank ([(0, n)] after ``gather`` / ``all_reduce``)."""
    _require_f32_sums(fusion, "merge_volumes")
    if getattr(fusion, "_shard_stripes", None) is not None:
        raise SafError("this volume was already merged and holds only its voxel stripes")
    flush = getattr(fusion, "flush", None)
    if flush is not None:
        flush()  # frames still queued behind integrate() belong to this job
    if fusion.accum_mode != _abi.SAF_SUM:
        means_to_sums(fusion)
    tensors = _volume_tensors(fusion)
    n = fusion.tsdf.numel()
    plans = []
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    # (the packed route is probed -- once per process -- before the first collective touches the volume)
    wish = sparse_wish(sparse, fusion.tsdf.device, group) if world > 1 and mode == "reduce_scatter" else 0.0
    try:
        stripes = merge_sums(tensors, group=group, mode=mode, gather=False, piece_bytes=piece_bytes, sparse=wish, plans_out=plans)
        for first, count in stripes:
            finalize_sums(fusion, first, count)
    except BaseException:
        # weight is all-reduced, some pieces are summed, others not: neither this rank's sums nor the job's.  Unusable until
        # reset(), like a failed flush -- a retry would add the other ranks' rows twice.
        fusion.__dict__["_poisoned"] = "merge_volumes failed part-way: the volume is half merged"
        raise
    if stripes == [(0, n)]:
        return stripes
    # voxel-sharded result: only the stripes hold the job's means; the other rows hold this rank's partial sums.
    # integrate() refuses to fuse into it (clipfusion._fuse_now) until gather_shards() has made it whole.
    fusion.accum_mode = _abi.SAF_RUNNING_MEAN
    fusion._shard_stripes = list(stripes)
    fusion.__dict__["_shard_plans"] = plans
    if gather:
        gather_shards(fusion, group)
        return [(0, n)]
    return stripes


def gather_shards(fusion, group=None):
    """All-gather a voxel-sharded merged volume (``merge_volumes(..., gather=False)`` or ``fuse_merge_pipelined``) so that every
    rank holds the whole merged volume and may fuse further frames.  Along the stripe plans the merge recorded
    (``_shard_plans``: one per merge, or one per slab of the pipelined merge) this is the in-place striped all-gather -- one
    ``all_gather_into_tensor`` per piece and tensor, every link busy.  A volume whose stripes were set by hand (no plan) falls
    back to exchanging the stripe lists (a few integers) and one rooted broadcast per stripe, tensor and 1 GiB chunk."""
    n = fusion.tsdf.numel()
    mine = getattr(fusion, "_shard_stripes", None)
    if mine is None:
        return [(0, n)]
    world = dist.get_world_size(group)
    tensors = _volume_tensors(fusion)
    plans = getattr(fusion, "_shard_plans", None)
    if plans:
        rank = dist.get_rank(group)
        for plan in plans:
            for t in tensors.values():
                _all_gather_striped(t, plan, group, rank, world)
    else:
        every = [None] * world
        dist.all_gather_object(every, [tuple(int(v) for v in st) for st in mine], group=group)
        for k, stripes in enumerate(every):
            for first, count in stripes:
                for t in tensors.values():
                    for c in _chunks(t[first : first + count]):
                        dist.broadcast(c, src=_global_rank(group, k), group=group)
    fusion._shard_stripes = None
    fusion.__dict__["_shard_plans"] = None
    return [(0, n)]


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


def _stripes_for_query(fusion, n, rank, world):
    st = getattr(fusion, "_shard_stripes", None)
    return ([tuple(x) for x in st], True) if st is not None else ([voxel_shard(n, rank, world)], False)


def query_sharded(fusion, text, epilogue="row_argmax", group=None, gather=True, dtype=torch.float16, scan_fn=None, **kw):
    """BASELINE config 5: many text queries over the merged volume, voxel-sharded over the ranks.

    After ``merge_volumes(..., gather=False)`` / ``fuse_merge_pipelined`` rank k holds the means of its STRIPES
    (``_shard_stripes``: one per piece of the merge); a volume that is whole on every rank is split by ``voxel_shard``.
    Each rank scans ONLY its stripes with ``query_scan_wide`` (no data-path collective: the volume never moves) and the
    small results are combined:

    * ``"query_max"``   -> (value [Q], voxel [Q]) identical on every rank: an all-gather of Q (score, voxel) pairs and a
      max -- equal scores resolve to the smaller voxel index, as a single-rank scan would;
    * ``"row_argmax"``  -> (index, value) of this rank's stripes (in stripe order), or of all N voxels in voxel order when
      ``gather`` (all-gather of 4 + 4 bytes per voxel);
    * ``"scores"`` / ``"vs_background"`` -> this rank's [count, Q] rows, stripe after stripe (33 GB in all at config 5: it
      stays sharded).

    ``scan_fn(feats, text, epilogue, row_offset=..., **kw)`` defaults to the HIP scan; the CPU tests inject an
    oracle-backed one to exercise the sharding and the reductions under gloo."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = fusion.tsdf.numel()
    stripes, from_merge = _stripes_for_query(fusion, n, rank, world)
    hip = scan_fn is None
    if hip:
        from .clipfusion import query_scan_wide

        scan_fn = query_scan_wide
    parts = []
    for first, count in stripes:
        feats = shard_features_16(fusion, first, count, dtype) if hip else fusion.clip_feat[first:first + count]
        parts.append(scan_fn(feats, text, epilogue, row_offset=first, **kw))
    big = torch.iinfo(torch.int64).max

    def best_of(vals, rows):  # [k, Q] candidates -> the maximum; among equals the smallest voxel index
        best = vals.max(dim=0).values
        cand = torch.where((vals == best[None]) & (rows >= 0), rows, torch.full_like(rows, big))
        pick = cand.min(dim=0).values
        return best, torch.where(pick == big, torch.full_like(pick, -1), pick)

    if epilogue == "query_max":
        val, row = parts[0] if len(parts) == 1 else best_of(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
        if world == 1:
            return val, row
        vals = [torch.empty_like(val) for _ in range(world)]
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(vals, val.contiguous(), group=group)
        dist.all_gather(rows, row.contiguous(), group=group)
        return best_of(torch.stack(vals), torch.stack(rows))
    if isinstance(parts[0], tuple):
        res = tuple(torch.cat([p[i] for p in parts]) for i in range(len(parts[0]))) if len(parts) > 1 else parts[0]
    else:
        res = torch.cat(parts) if len(parts) > 1 else parts[0]
    if world == 1:
        return res
    if epilogue == "row_argmax" and gather:
        idx, val = res
        every = [None] * world
        dist.all_gather_object(every, stripes, group=group)
        counts = [sum(c for _, c in st) for st in every]
        pad = max(counts)
        out = []
        for t in (idx, val):
            mine = torch.zeros(pad, dtype=t.dtype, device=t.device)
            mine[: t.numel()] = t
            got = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(got, mine, group=group)
            full = torch.empty(n, dtype=t.dtype, device=t.device)
            for k, st in enumerate(every):  # rank k's results lie stripe after stripe
                o = 0
                for first, count in st:
                    full[first : first + count] = got[k][o : o + count]
                    o += count
            out.append(full)
        return tuple(out)
    return res
