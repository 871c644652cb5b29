"""On the GPU box: config 5's matrix-valued scans with the output rows padded to 16 bytes (stride 1000) and to whole 128-byte
lines (stride 1024): how much of their time is partial-line write traffic."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatially_aware_ai_amd.clipfusion import query_scan_wide

dev = "cuda:0"
n, d, q, n_bg = 256 ** 3, 512, 1000, 4
g = torch.Generator(device=dev).manual_seed(100)
feats = torch.empty((n, d), dtype=torch.float16, device=dev)
for s0 in range(0, n, 1 << 20):
    feats[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), d), generator=g, device=dev).half()
text = torch.randn((n_bg + q, d), generator=torch.Generator().manual_seed(9))
text = (text / text.norm(dim=-1, keepdim=True)).to(dev)
for stride in (1000, 1024, 1000, 1024):
    big = torch.empty((n, stride), dtype=torch.float16, device=dev)[:, :q]
    for name, fn in (("heat maps", lambda: query_scan_wide(feats, text, "vs_background", scale=100.0, n_background=n_bg, rescale=True, out=big)),
                     ("raw scores", lambda: query_scan_wide(feats, text[n_bg:], "scores", out=big))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        print(f"out stride {stride}: {name} {e0.elapsed_time(e1) / 2:.2f} ms", flush=True)
    del big
