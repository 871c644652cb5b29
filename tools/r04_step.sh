#!/bin/bash
# round 4, on the GPU box: windows cut into slabs (SAF_WIN_SLABS / SAF_WIN_W0_SLABS), and the per-phase wave cycles of the row kernel
line() { python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']
    print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows/window', r['avg_launch_us'], 'frac', r['frac'], 'alone', (r.get('isolated') or {}).get('avg_launch_us'), (r.get('isolated') or {}).get('frac'), 'classify', d['kernel_breakdown']['sweep_us'])
except Exception as e:
    print('$1 FAILED', e, t[-3:])"; }
B="python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 5 --warmup 2"
for k in A B; do for env in ${ENVS:-SAF_X=1 SAF_WIN_SLABS=2 SAF_WIN_SLABS=4 SAF_WIN_W0_SLABS=4}; do env $env timeout -k 10 200 $B --depth-kind $k 2>/dev/null | line "sums $env depth=$k"; done; done
SAF_WIN_FORM=rows timeout -k 10 200 $B 2>/dev/null | line "rows depth=A"
if [ -z "$NO_TIMING" ]; then
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -DSAF_WIN_TIMING" > /dev/null 2>&1)
for k in A B; do SAF_WIN_OVERLAP=0 timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 1 --warmup 0 --depth-kind $k 2>&1 >/dev/null | grep "win timing" | tail -2 | sed "s/^/depth $k: /"; done
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make > /dev/null 2>&1)
fi
