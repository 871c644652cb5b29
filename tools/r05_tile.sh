#!/bin/bash
# round 5: shapes of the classification's depth tiles (SAF_CLS_TILE_WL2: 2 = 4 x 8 pixels, 3 = 8 x 4, 4 = 16 x 2), same box
OUT=${1:-gpurun_out/r05tile}; mkdir -p $OUT
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v saf_window.o | tr "\n" " ")
for v in ${VARIANTS:-3 2 4}; do
  /opt/rocm/bin/hipcc $FLAGS -DSAF_CLS_TILE_WL2=$v -c $C/saf_window.hip -o /tmp/win_t$v.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_t$v.so /tmp/win_t$v.o $OTHERS || continue
  for k in ${KINDS:-A B}; do for rep in 1 2; do
  SAF_LIB_PATH=/tmp/libsaf_t$v.so python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-side --end-to-end 0 --no-pmc --depth-kind $k 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('tile width 2^$v depth $k:', j['value'], 'frames/s', j['ms_per_step'], 'ms  classify', j['kernel_breakdown']['sweep_us'], 'us  rows', j['kernel_breakdown']['fuse_us'], 'us  frac', r['frac'], ' alone: rows', r['isolated']['avg_launch_us'], 'classify', r['isolated'].get('classify_us_alone'))"
  done; done
done
