#!/bin/bash
# round 5: scene B with 2 / 3 / 4 row-kernel workgroups per CU (SAF_WIN_WGS), HITCAP=256 build (32 KB of LDS per workgroup)
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v saf_window.o | tr "\n" " ")
for cap in 256 128; do
/opt/rocm/bin/hipcc $FLAGS -DSAF_WIN_HITCAP=$cap -c $C/saf_window.hip -o /tmp/win_h.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_h.so /tmp/win_h.o $OTHERS || exit 1
for k in B A; do for w in 2 3 4; do for ov in 1 0; do
  SAF_WIN_WGS=$w SAF_WIN_OVERLAP=$ov SAF_LIB_PATH=/tmp/libsaf_h.so python3 bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-side --end-to-end 0 --no-pmc --depth-kind $k 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HITCAP=$cap depth $k wgs/CU $w overlap=$ov', j['value'], 'frames/s', j['ms_per_step'], 'ms  classify', j['kernel_breakdown']['sweep_us'], 'us  rows', j['kernel_breakdown']['fuse_us'], 'us')"
done; done; done; done
