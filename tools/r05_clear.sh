#!/bin/bash
# round 5, on the GPU box: the deferred clear of a recycled volume beside the last window's row kernel (saf_fuse_frames_recycled)
# against the same zeros written behind it (SAF_WIN_CLEAR_BESIDE=0), same box, depth A and the coherent scene B;
# SAF_CLEAR_WGS = workgroups per CU of the clear kernel (default 4 beside a row kernel, 16 alone).
OUT=${1:-gpurun_out/r05clear}
mkdir -p $OUT
B="python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --no-pmc --steps 10 --warmup 3"
for k in ${KINDS:-B A}; do
  for v in ${VARS:-0 1:2 1:4 1:8 1:16 0 1:4}; do
    b=${v%%:*}; w=${v##*:}; [ "$w" = "$v" ] && w=""
    SAF_WIN_CLEAR_BESIDE=$b SAF_CLEAR_WGS=$w timeout -k 10 200 $B --depth-kind $k 2>$OUT/${k}_$v.err > $OUT/${k}_$v.json && python3 -c "
import json
d=json.loads(open('$OUT/${k}_$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('depth $k beside=$b wgs/CU=${w:-default}', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows/window', r['avg_launch_us'], 'frac', r['frac'])"
  done
done 2>&1 | tee $OUT/clear_beside.txt
