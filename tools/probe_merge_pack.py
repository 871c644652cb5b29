#!/usr/bin/env python3
"""Rates of the packed merge's device passes (csrc/saf_merge.hip) against the torch passes they replace (round 5:
nonzero + index_select, sum over the stacked contributions, index_copy_), one GPU, no collective: a slab of 2 M rows x 512 f32
(4.3 GB, 1/8 of BASELINE's volume), `share` of the rows touched, world = 8 (this rank sends share x 2 M rows and adds 8
contributions of its own eighth).  Prints ms and ms per GB moved (pack: read + write of the touched rows; add: world reads +
one write).  python tools/probe_merge_pack.py [share]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from spatially_aware_ai_amd._lib import check, lib  # noqa: E402

share = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
L = lib()
dev = torch.device("cuda", 0)
n, d, world = 1 << 21, 512, 8
g = torch.Generator(device=dev).manual_seed(1)
# a coherent scene's touched rows come in runs (a shell crossing z-columns): runs of 8 rows
w = (torch.rand(n // 8, generator=g, device=dev) < share).int().repeat_interleave(8)
t = torch.randn((n, d), generator=g, device=dev)
c = n // world
s = torch.cuda.current_stream().cuda_stream
bounds = torch.tensor([k * c for k in range(world + 1)], dtype=torch.int64, device=dev)
pos = torch.empty(n + 1, dtype=torch.int32, device=dev)
ws = torch.empty(L.saf_merge_scan_workspace_bytes(n, world + 1), dtype=torch.uint8, device=dev)
host = torch.empty(world + 1, dtype=torch.int32).pin_memory()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ms_scan = timed(lambda: check(L.saf_merge_scan_touched(w.data_ptr(), n, pos.data_ptr(), bounds.data_ptr(), world + 1, host.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), s), "scan"))
torch.cuda.synchronize()
offs = host.tolist()
total, mine = offs[world], offs[world] - offs[world - 1]
send = torch.empty((total, d), device=dev)
recv = torch.randn((world * mine, d), generator=g, device=dev)
gb_pack = 2 * total * d * 4 / 1e9
gb_add = (world + 1) * mine * d * 4 / 1e9
ms_pack = timed(lambda: check(L.saf_merge_pack_rows(t.data_ptr(), d * 4, w.data_ptr(), pos.data_ptr(), 0, n, send.data_ptr(), s), "pack"))
ms_add = timed(lambda: check(L.saf_merge_add_packed(t.data_ptr(), d * 4, 1, w.data_ptr(), pos.data_ptr(), (world - 1) * c, c, recv.data_ptr(),
                                                     mine, world, s), "add"))
idx_box = {}


def torch_idx():
    touched = w > 0
    cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(touched, 0, dtype=torch.int64)])
    cs[bounds].cpu().tolist()
    idx_box["idx"] = torch.nonzero(touched).squeeze(1)


ms_tidx = timed(torch_idx)
idx = idx_box["idx"]
ms_tpack = timed(lambda: t.index_select(0, idx))
my_idx = idx[offs[world - 1]:offs[world]]
ms_tadd = timed(lambda: t.index_copy_(0, my_idx, recv.view(world, mine, d).sum(dim=0)))
print(f"slab {n} rows x {d} f32, {share:.0%} touched ({total} rows sent, {mine} rows owned), world {world}")
print(f"  HIP  : scan {ms_scan:.3f} ms | pack {ms_pack:.3f} ms = {ms_pack / gb_pack:.3f} ms/GB ({gb_pack / ms_pack:.2f} TB/s) | "
      f"add {ms_add:.3f} ms = {ms_add / gb_add:.3f} ms/GB ({gb_add / ms_add:.2f} TB/s) | pack + add {(ms_pack + ms_add) / (gb_pack + gb_add):.3f} ms per GB moved")
print(f"  torch: cumsum + nonzero + sync {ms_tidx:.3f} ms | index_select {ms_tpack:.3f} ms = {ms_tpack / gb_pack:.3f} ms/GB | "
      f"sum + index_copy_ {ms_tadd:.3f} ms = {ms_tadd / gb_add:.3f} ms/GB | pack + add {(ms_tpack + ms_tadd) / (gb_pack + gb_add):.3f} ms per GB moved")
