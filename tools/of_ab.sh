#!/bin/bash
# development A/B of the order-free row kernel's compile-time options on one box:
#   bash tools/of_ab.sh "<-D flags>[;ENV=..]" ...       (first line: the build as it is)
run() { for k in ${KINDS:-A B}; do env SAF_WIN_FORM=sums $2 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 3 --warmup 1 --depth-kind $k 2>/dev/null | python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']
    print('$1 $2', 'depth $k:', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows', r['avg_launch_us'], 'alone', (r.get('isolated') or {}).get('avg_launch_us'), 'classify', d['kernel_breakdown']['sweep_us'])
except Exception as e:
    print('$1 $2 depth $k FAILED', e)"; done; }
run "default" "SAF_X=1"
for spec in "$@"; do
  f="${spec%%;*}"; e="${spec#*;}"; [ "$e" = "$spec" ] && e="SAF_X=1"
  (cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $f" > /dev/null 2>&1) && run "$f" "$e"
done
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make > /dev/null 2>&1)
