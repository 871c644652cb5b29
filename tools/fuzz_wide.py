"""On the GPU box: differential check of the two forms of the fused wide scan (v_mfma 16x16x32, the default, against 32x32x16,
SAF_WIDE_MFMA=32) over random shapes: scores / heat maps agree to rounding, the reductions pick (nearly) equal maxima."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatially_aware_ai_amd.clipfusion import query_scan_wide

g = torch.Generator().manual_seed(7)
bad = 0
for it in range(40):
    d = [256, 512][int(torch.randint(0, 2, (1,), generator=g))]
    dt = [torch.float16, torch.bfloat16][int(torch.randint(0, 2, (1,), generator=g))]
    odt = [torch.float16, torch.bfloat16, torch.float32][int(torch.randint(0, 3, (1,), generator=g))]
    n = int(torch.randint(1, 200000, (1,), generator=g)) if it % 4 else int(torch.randint(60000, 400000, (1,), generator=g))
    q = int(torch.randint(1, 700, (1,), generator=g))
    n_bg = int(torch.randint(1, min(q, 32) + 1, (1,), generator=g)) if q > 1 else 0
    feats = torch.randn(n, d, generator=g).to(dt).cuda()
    text = torch.randn(q, d, generator=g)
    text = (text / text.norm(dim=-1, keepdim=True)).cuda()
    out = {}
    for form in ("16", "32"):
        os.environ["SAF_WIDE_MFMA"] = form
        r = {"scores": query_scan_wide(feats, text, "scores", out_dtype=odt).float()}
        if n_bg and q - n_bg >= 1:
            r["vsbg"] = query_scan_wide(feats, text, "vs_background", scale=100.0, n_background=n_bg, rescale=bool(it & 1), out_dtype=odt).float()
        r["ra"] = query_scan_wide(feats, text, "row_argmax")
        r["qm"] = query_scan_wide(feats, text, "query_max", row_offset=it)
        out[form] = r
    tol = {torch.float32: 2e-5, torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}[odt]
    a, b = out["16"], out["32"]
    e_s = (a["scores"] - b["scores"]).abs().max().item()
    e_v = (a["vsbg"] - b["vsbg"]).abs().max().item() if "vsbg" in a else 0.0
    e_ra = (a["ra"][1] - b["ra"][1]).abs().max().item()
    agree_ra = (a["ra"][0] == b["ra"][0]).float().mean().item()
    e_qm = (a["qm"][0] - b["qm"][0]).abs().max().item()
    agree_qm = (a["qm"][1] == b["qm"][1]).float().mean().item()
    ok = e_s <= tol and e_v <= max(tol, 3e-3) and e_ra <= 2e-5 and e_qm <= 2e-5 and agree_ra > 0.995 and agree_qm > 0.97
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} it={it} n={n} q={q} d={d} {str(dt)[6:]}->{str(odt)[6:]} n_bg={n_bg}: scores {e_s:.2e} vsbg {e_v:.2e} argmax val {e_ra:.1e} agree {agree_ra:.4f} qmax val {e_qm:.1e} agree {agree_qm:.3f}", flush=True)
print("FAILED" if bad else "all agree", bad)
sys.exit(1 if bad else 0)
