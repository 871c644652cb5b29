#!/usr/bin/env python3
"""Development: where a step of query_wide2_kernel spends its cycles (build the library with -DSAF_W2_STAMP first:
make -C spatially_aware_ai_amd/csrc HIPFLAGS="... -DSAF_W2_STAMP").  Runs config 5's scans once each and prints, per wave of
workgroup 0, the s_memtime cycles of every segment of the step loop (tools/w2_stamps.sh does both)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatially_aware_ai_amd import _lib
from spatially_aware_ai_amd.clipfusion import query_scan_wide

lib = _lib.lib()
fn = lib.saf_debug_w2_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
fn.restype = ctypes.c_int
dev = torch.device("cuda", 0)
n, d, q, n_bg = 256 ** 3, 512, 1000, 4
g = torch.Generator(device=dev).manual_seed(100)
feats = torch.empty((n, d), dtype=torch.float16, device=dev)
for s0 in range(0, n, 1 << 20):
    feats[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), d), generator=g, device=dev).half()
text = torch.randn((n_bg + q, d), generator=torch.Generator().manual_seed(9))
text = (text / text.norm(dim=-1, keepdim=True)).to(dev)
big = torch.empty((n, 1000), dtype=torch.float16, device=dev)
names7 = "wait for the tile (vmcnt)"
names = ["tail of the step before", "new block: row loads + norms", "wait for the tile + barrier", "issue of the next tile's DMA",
         "first tile of a block (waits for its rows)", "tile: LDS reads, MFMAs, epilogue pieces"]
cases = {"row_argmax": lambda: query_scan_wide(feats, text[n_bg:], "row_argmax"),
         "heat_maps": lambda: query_scan_wide(feats, text, "vs_background", scale=100.0, n_background=n_bg, rescale=True, out=big),
         "query_max": lambda: query_scan_wide(feats, text[n_bg:], "query_max")}
for name, f in cases.items():
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert fn(buf) == 0
    print(f"{name}: {e0.elapsed_time(e1):.2f} ms (instrumented)")
    for w in range(8):
        v = list(buf[w * 8:w * 8 + 8])
        if v[6] == 0:
            continue
        tot = sum(v[:6]) + v[7]
        print(f"  wave {w}: steps {v[6]}, cycles/step {tot / v[6]:.0f} | " + " | ".join(f"{names[k][:22]} {v[k] / v[6]:.0f}" for k in range(6)) + f" | vmcnt wait {v[7] / v[6]:.0f}")
