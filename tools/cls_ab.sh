#!/bin/bash
# A/B of classification build options on one box: bash tools/cls_ab.sh "<-D flags>" ...   (classification alone + the job, depth A and B)
run() { for k in A B; do timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 4 --warmup 2 --depth-kind $k 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 depth $k:', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows', r['avg_launch_us'], 'classify beside', d['kernel_breakdown']['sweep_us'], 'alone', r['isolated']['classify_us_alone'])"; done; }
run "default"
for f in "$@"; do
  (cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $f" > /dev/null 2>&1) && run "$f"
done
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make > /dev/null 2>&1)
