#!/usr/bin/env python3
"""The split scan's time by epilogue and by what it writes: L labels over `rows` x 512 fp32 rows, raw scores / softmax / surgery,
the whole [N, L] matrix or the last column only (same reads, same MFMAs).  python tools/probe_qsplit_writes.py [rows = 2^23] [L ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from spatially_aware_ai_amd import _abi  # noqa: E402
from spatially_aware_ai_amd.clipfusion import _query_scan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
ls = [int(x) for x in sys.argv[2:]] or [63, 64, 33, 32, 5]
d = 512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
f32 = torch.empty((n, d), device=dev)
for s0 in range(0, n, 1 << 20):
    f32[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), d), generator=g, device=dev)
for nl in ls:
    t = torch.nn.functional.normalize(torch.randn((nl, d), generator=g, device=dev), dim=-1)
    for name, epi, scale in (("scores", _abi.SAF_Q_SCORES, 1.0), ("softmax", _abi.SAF_Q_SOFTMAX, 100.0), ("surgery", _abi.SAF_Q_SURGERY, 1.0)):
        res = []
        for last in (False, True):
            fn = lambda: _query_scan(f32, t, epi, scale=scale, normalize=True, last_only=last)
            fn(); fn()
            torch.cuda.synchronize()
            ms = 1e9
            for _rep in range(3):  # best of three timings of five scans
                t0 = time.perf_counter()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                ms = min(ms, (time.perf_counter() - t0) / 5 * 1e3)
            nbytes = n * d * 4 + n * (1 if last else nl) * 4
            res.append(f"{'last column' if last else 'matrix'} {ms:.3f} ms = {nbytes / ms / 1e6:.0f} GB/s")
        print(f"L={nl} {name}: " + " | ".join(res), flush=True)
