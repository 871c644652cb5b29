#!/usr/bin/env python3
"""L = 63 surgery over 8 GB of fp32 rows of 768 / 1024 channels: the matrix-core scan 32 labels at a time (the whole matrix) against
the one-wave-per-row kernel (what the same call took in rounds 1-5; still what a last-column-only call of this shape takes)."""
import sys, time, torch
sys.path.insert(0, ".")
from spatially_aware_ai_amd import _abi
from spatially_aware_ai_amd.clipfusion import _query_scan
dev = torch.device("cuda", 0)
for d in (768, 1024):
    n = (1 << 33) // (d * 4)  # 8 GB of rows
    f = torch.randn((n, d), device=dev)
    t = torch.nn.functional.normalize(torch.randn((63, d), device=dev), dim=-1)
    fn = lambda: _query_scan(f, t, _abi.SAF_Q_SURGERY, scale=1.0, normalize=True)
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    fl = lambda: _query_scan(f, t, _abi.SAF_Q_SURGERY, scale=1.0, normalize=True, last_only=True)
    fl(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fl()
    torch.cuda.synchronize()
    ms1 = (time.perf_counter() - t0) * 1e3
    print(f"D={d} L=63 surgery over {n} rows: blocks of 32 on the matrix cores {ms:.2f} ms ({n * d * 4 / ms / 1e6:.0f} GB/s of rows) | one wave per row {ms1:.2f} ms", flush=True)
    del f
