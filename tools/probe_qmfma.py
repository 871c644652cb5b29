#!/usr/bin/env python3
"""The reference's own scans over the fp32 volume (query_split_kernel, or query_mfma_kernel under SAF_Q_SPLIT=0: L = 5 softmax, query_mesh.py:36-39; L = 63 surgery,
:52-83) alone, over `rows` x 512 fp32 rows: ms per scan, HBM GB/s of the algorithmic bytes, exact-fp32 TFLOP/s.  For same-box
A/Bs of library builds through SAF_LIB_PATH (tools/build_variant.sh).  python tools/probe_qmfma.py [rows = 2^23] [L ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from spatially_aware_ai_amd import _abi  # noqa: E402
from spatially_aware_ai_amd.clipfusion import _query_scan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
ls = [int(x) for x in sys.argv[2:]] or [5, 63]
d = 512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
f32 = torch.empty((n, d), device=dev)
for s0 in range(0, n, 1 << 20):
    f32[s0:s0 + (1 << 20)] = torch.randn((min(1 << 20, n - s0), d), generator=g, device=dev)
out = []
for nl in ls:
    t = torch.nn.functional.normalize(torch.randn((nl, d), generator=g, device=dev), dim=-1)
    epi, scale, last = (_abi.SAF_Q_SOFTMAX, 100.0, True) if nl <= 8 else (_abi.SAF_Q_SURGERY, 1.0, False)
    fn = lambda: _query_scan(f32, t, epi, scale=scale, normalize=True, last_only=last)
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    nbytes = n * d * 4 + n * (1 if last else nl) * 4
    out.append(f"L={nl}: {ms:.3f} ms ({ms * (1 << 24) / n:.2f} at 256^3), {nbytes / ms / 1e6:.0f} GB/s, {2.0 * n * d * nl / ms / 1e9:.1f} TF")
print(" | ".join(out))
