import sys, time, torch
sys.path.insert(0, ".")
from spatially_aware_ai_amd import _abi
from spatially_aware_ai_amd.clipfusion import _query_scan
dev = torch.device("cuda", 0)
n, d = 1 << 24, 512
for dt in (torch.bfloat16, torch.float16):
    f = torch.empty((n, d), dtype=dt, device=dev)
    for s0 in range(0, n, 1 << 20):
        f[s0:s0 + (1 << 20)] = torch.randn((1 << 20, d), device=dev).to(dt)
    for nl, epi, scale, last in ((5, _abi.SAF_Q_SOFTMAX, 100.0, True), (63, _abi.SAF_Q_SURGERY, 1.0, False)):
        t = torch.nn.functional.normalize(torch.randn((nl, d), device=dev), dim=-1)
        fn = lambda: _query_scan(f, t, epi, scale=scale, normalize=True, last_only=last)
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        nb = n * d * 2 + n * (1 if last else nl) * 4
        print(f"{dt} L={nl}: {ms:.2f} ms = {nb / ms / 1e6:.0f} GB/s", flush=True)
    del f
