#!/bin/bash
# round 4: what BASELINE config 3's fused part (bf16 volume + label histogram) costs, piece by piece, against the f32 headline
line() { python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']
    print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows/window', r['avg_launch_us'], 'frac', r['frac'], 'alone', (r.get('isolated') or {}).get('avg_launch_us'), (r.get('isolated') or {}).get('frac'), 'classify', d['kernel_breakdown']['sweep_us'])
except Exception as e:
    print('$1 FAILED', e, t[-3:])"; }
B="python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 4 --warmup 2"
timeout -k 10 200 $B 2>/dev/null | line "f32"
timeout -k 10 200 $B --labels 2>/dev/null | line "f32 + labels"
timeout -k 10 200 $B --feat-dtype bf16 2>/dev/null | line "bf16"
timeout -k 10 200 $B --feat-dtype bf16 --labels 2>/dev/null | line "bf16 + labels (config 3)"
SAF_WIN_FORM=rows timeout -k 10 200 $B --feat-dtype bf16 --labels 2>/dev/null | line "bf16 + labels, frame-ordered rows (round 3)"
