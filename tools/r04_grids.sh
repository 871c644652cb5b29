#!/bin/bash
# round 4: the reference's grids (voxel_grid_compare.md) on the windowed path, D = 512 f32, against 128^3; per voxel-frame rates
for g in 128 127,104,116 118,115,113 61,60,59 57,56,55; do
  timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 5 --warmup 2 --grid $g 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; g=[int(x) for x in '$g'.split(',')]; g=g*3 if len(g)==1 else g
n=g[0]*g[1]*g[2]
print('grid $g: %8.1f frames/s  %7.3f ms per 512 frames  %6.2f G voxel-frames/s  rows/window %7.1f us  classify/launch %6.1f us' % (d['value'], d['ms_per_step'], d['value']*n/1e9, r['avg_launch_us'], d['kernel_breakdown']['sweep_us']))"
done
