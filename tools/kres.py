#!/usr/bin/env python3
"""Compact per-kernel resource table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, AGPRs, spilled
registers, scratch bytes, occupancy, LDS.  Usage: python tools/kres.py spatially_aware_ai_amd/csrc/saf_window.hip [name filter] [-- extra hipcc flags]"""
import re, subprocess, sys, os
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src = args[0]; flt = args[1] if len(args) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True, cwd=os.getcwd()).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("saf::(anonymous namespace)::", "").replace("void ", "")
        dem = re.sub(r"\(.*", "", dem)
        cur = {"name": dem}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    if flt and flt not in r["name"]: continue
    print(f"{r['name'][:60]:60s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
