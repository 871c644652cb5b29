#!/bin/bash
# A/B of the row kernel's forms on one box (frame-ordered rows / order-free sums), depth A and B: bash tools/sums_ab.sh [extra bench args]
line() { python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']
    print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows', r['avg_launch_us'], 'frac', r['frac'], 'alone', (r.get('isolated') or {}).get('avg_launch_us'), (r.get('isolated') or {}).get('frac'), 'classify', d['kernel_breakdown']['sweep_us'])
except Exception as e:
    print('$1 FAILED', e, t[-3:])"; }
for form in ${FORMS:-rows sums}; do for k in A B; do
  SAF_WIN_FORM=$form timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 3 --warmup 1 --depth-kind $k "$@" 2>gpurun_out/sums_ab_$form$k.err | line "form=$form depth=$k"
done; done
