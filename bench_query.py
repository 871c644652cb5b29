#!/usr/bin/env python3
"""Query-scan benchmark (SURVEY.md §8d 'query scan reported separately'): the CLIP-text scan over
a fused feature volume resident in HBM.

  python bench_query.py [--grid 256] [--dim 512] [--labels 5] [--epilogue softmax|surgery|scores]
                        [--dtype f32|bf16|f16] [--last-only]

Prints one JSON line: rows/s, achieved HBM GB/s against the algorithmic bytes
N*D*s + Q*D*4 + N*Q_out*4 and, for large Q, TFLOP/s (2*N*D*Q)."""
import argparse
import json
import time

import torch

from spatially_aware_ai_amd import _abi
from spatially_aware_ai_amd.clipfusion import _query_scan, query_scores_wide


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--labels", type=int, default=5)
    ap.add_argument("--epilogue", default="softmax", choices=["softmax", "surgery", "scores"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--last-only", action="store_true")
    ap.add_argument("--rows", type=int, default=0, help="override the number of rows (default grid^3)")
    ap.add_argument("--wide", action="store_true",
                    help="config 5: many queries over a 16-bit volume on the 16-bit matrix cores (scores, 16-bit out)")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = a.rows or a.grid**3
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    g = torch.Generator(device=dev).manual_seed(0)
    feats = torch.empty((n, a.dim), dtype=dt, device=dev)
    step = 1 << 20
    for s in range(0, n, step):  # fill in slabs: randn in f32 then cast
        feats[s : s + step] = torch.randn((min(step, n - s), a.dim), generator=g, device=dev).to(dt)
    text = torch.randn((a.labels, a.dim), generator=g, device=dev)
    text = text / text.norm(dim=-1, keepdim=True)
    epi = {"softmax": _abi.SAF_Q_SOFTMAX, "surgery": _abi.SAF_Q_SURGERY, "scores": _abi.SAF_Q_SCORES}[a.epilogue]
    scale = 100.0 if a.epilogue == "softmax" else 1.0
    run = lambda: _query_scan(feats, text, epi, scale=scale, normalize=True, last_only=a.last_only)
    if a.wide:
        run = lambda: query_scores_wide(feats, text, scale=1.0, normalize=True)
    run()
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = run()
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / a.steps
    esz = feats.element_size()
    q_out = 1 if a.last_only else a.labels
    nbytes = n * a.dim * esz + a.labels * a.dim * 4 + n * q_out * (esz if a.wide else 4)
    flops = 2.0 * n * a.dim * a.labels
    print(json.dumps({
        "metric": "query scan rows/s", "value": round(n / dt_s, 1), "ms": round(dt_s * 1e3, 3), "rows": n,
        "dim": a.dim, "labels": a.labels, "epilogue": a.epilogue, "dtype": a.dtype, "last_only": a.last_only,
        "hbm_GBps": round(nbytes / dt_s / 1e9, 1), "hbm_frac_of_8TBps": round(nbytes / dt_s / 8e12, 4),
        "tflops": round(flops / dt_s / 1e12, 2), "wide": a.wide,
        "mfma_frac_of_2.5PF": round(flops / dt_s / 2.5e15, 4) if a.wide else None,
    }))


if __name__ == "__main__":
    main()
