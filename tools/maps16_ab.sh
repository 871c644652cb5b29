#!/bin/bash
# A/B of the bf16 map images (bf16 volumes, order-free form) on one box: bash tools/maps16_ab.sh ["-D flags" ...]
line() { python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']
    print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows/window', r['avg_launch_us'], 'frac', r['frac'], 'alone', (r.get('isolated') or {}).get('avg_launch_us'), 'classify', d['kernel_breakdown']['sweep_us'])
except Exception as e:
    print('$1 FAILED', e, t[-3:])"; }
B="python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 4 --warmup 2 --grid 256 --feat-dtype bf16"
run() { timeout -k 10 200 $B 2>/dev/null | line "$1 bf16 256^3"; timeout -k 10 200 $B --labels 2>/dev/null | line "$1 bf16 256^3 + labels (config 3)"; timeout -k 10 200 $B --labels --depth-kind B 2>/dev/null | line "$1 config 3, scene B"; }
run "default"
for f in "$@"; do
  (cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $f" > /dev/null 2>&1) && run "$f"
done
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make > /dev/null 2>&1)
