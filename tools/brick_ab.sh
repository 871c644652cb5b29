#!/bin/bash
# development A/B of a saf_brick.hip compile-time option on one box: bash tools/brick_ab.sh "<-D flags>" ...
run() { for k in A B; do SAF_WIN_FORM=bricks timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 2 --warmup 1 --depth-kind $k 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'depth $k:', d['value'], 'frames/s', d['ms_per_step'], 'ms; walk', r['avg_launch_us'], 'alone', (r.get('isolated') or {}).get('avg_launch_us'))"; done; }
run "default"
for f in "$@"; do
  (cd spatially_aware_ai_amd/csrc && touch saf_brick.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $f" > /dev/null 2>&1) && run "$f"
done
