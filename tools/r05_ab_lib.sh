#!/bin/bash
# same-box A/B of the shipped library against another build of it (tools/_ab/libsaf_old.so: a worktree of an earlier revision, built
# in the container): bash tools/r05_ab_lib.sh "<bench args>" ...   -- one line per (args, build, repeat)
for args in "$@"; do for rep in 1 2; do for tag in old new; do
  lib=tools/_ab/libsaf_old.so; [ $tag = new ] && lib=spatially_aware_ai_amd/libsaf_hip.so
  SAF_LIB_PATH=$lib timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-side --end-to-end 0 --no-pmc $args 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('[$args] $tag', j['value'], 'frames/s', j['ms_per_step'], 'ms  classify', j['kernel_breakdown']['sweep_us'], 'us  rows', j['kernel_breakdown']['fuse_us'], 'us  alone', (r.get('isolated') or {}).get('avg_launch_us'), 'frac', r['frac'])"
done; done; done
