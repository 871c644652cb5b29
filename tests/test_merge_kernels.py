"""The device passes of the frame-sharded merge's packed route (csrc/saf_merge.hip; SURVEY section 8e, new capability):
`saf_merge_scan_touched` (positions of the touched rows + the split sizes), `saf_merge_pack_rows` (touched rows -> the send
buffer), `saf_merge_add_packed` (the received contributions added in rank order, in place) against their torch statements --
bit for bit: the sums are left-to-right in rank order on both sides.  The collectives around them are tested under gloo at
world 2 / 4 / 8 (tests/test_distributed_cpu.py) and under RCCL at world 1 (tests/test_nccl_world1.py)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,share,world", [(5000, 0.2, 4), (70001, 0.03, 8), (4096, 1.0, 2), (300, 0.0, 5)])
def test_scan_pack_add_against_torch(n, share, world):
    from spatially_aware_ai_amd._lib import check, lib

    L = lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(n)
    w = (torch.rand(n, generator=g, device=dev) < share).int() * torch.randint(1, 9, (n,), generator=g, device=dev, dtype=torch.int32)
    c = n // world
    bounds = torch.tensor([k * c for k in range(world + 1)], dtype=torch.int64, device=dev)
    pos = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ws = torch.empty(L.saf_merge_scan_workspace_bytes(n, world + 1), dtype=torch.uint8, device=dev)
    host = torch.empty(world + 1, dtype=torch.int32).pin_memory()
    s = torch.cuda.current_stream().cuda_stream
    check(L.saf_merge_scan_touched(w.data_ptr(), n, pos.data_ptr(), bounds.data_ptr(), world + 1, host.data_ptr(), ws.data_ptr(),
                                   ws.numel(), s), "scan")
    torch.cuda.synchronize()
    want = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(w > 0, 0)]).int()
    assert torch.equal(pos, want) and host.tolist() == want[bounds].tolist()
    idx = torch.nonzero(w > 0).squeeze(1)
    for shape, dt in (((n, 512), torch.float32), ((n, 3), torch.float32), ((n, 143), torch.int32), ((n,), torch.float32)):
        t = (torch.randn(shape, generator=g, device=dev) * 100).to(dt)
        row_bytes = (t[0].numel() if t.dim() > 1 else 1) * 4
        # pack: the rows of parts 1 .. world - 1 (a range that does not start at row 0)
        first, rows = c, (world - 1) * c
        sel = idx[(idx >= first) & (idx < first + rows)]
        send = torch.full((len(sel),) + tuple(shape[1:]), 7, dtype=dt, device=dev)
        if len(sel):  # (an empty send buffer has no address: distributed.py skips a piece nobody touched)
            check(L.saf_merge_pack_rows(t.data_ptr(), row_bytes, w.data_ptr(), pos.data_ptr(), first, rows, send.data_ptr(), s), "pack")
        assert torch.equal(send, t[sel])
        # add: this "rank" owns the last part; world contributions of its touched rows
        mine_rows = idx[(idx >= (world - 1) * c) & (idx < world * c)]
        mine = len(mine_rows)
        recv = (torch.randn((world * mine,) + tuple(shape[1:]), generator=g, device=dev) * 100).to(dt)
        dst = t.clone()
        check(L.saf_merge_add_packed(dst.data_ptr(), row_bytes, 1 if dt.is_floating_point else 0, w.data_ptr(), pos.data_ptr(),
                                     (world - 1) * c, c, recv.data_ptr(), mine, world, s), "add")
        ref = t.clone()
        if mine:
            parts = recv.view((world, mine) + tuple(shape[1:]))
            acc = parts[0].clone()
            for k in range(1, world):
                acc = acc + parts[k]
            ref[mine_rows] = acc
        assert torch.equal(dst, ref), (shape, dt)
