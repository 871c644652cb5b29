#!/bin/bash
# same-box A/B of the working tree's saf_window.hip against the one of a git revision (default HEAD): bash tools/r05_ab_git.sh [rev] -- needs the file
# tools/_ab_old_window.hip written by the caller (the GPU box has no .git): git show REV:spatially_aware_ai_amd/csrc/saf_window.hip > tools/_ab_old_window.hip
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function -I$C"
OTHERS=$(ls $C/*.o | grep -v saf_window.o | tr "\n" " ")
/opt/rocm/bin/hipcc $FLAGS -x hip -c tools/_ab_old_window.hip -o /tmp/win_old.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_old.so /tmp/win_old.o $OTHERS || exit 1
for k in ${KINDS:-B A}; do for rep in 1 2; do for tag in old new; do
  lib=/tmp/libsaf_old.so; [ $tag = new ] && lib=spatially_aware_ai_amd/libsaf_hip.so
  SAF_LIB_PATH=$lib python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-side --end-to-end 0 --no-pmc --depth-kind $k ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('depth $k $tag', j['value'], 'frames/s', j['ms_per_step'], 'ms  classify', j['kernel_breakdown']['sweep_us'], 'us  rows', j['kernel_breakdown']['fuse_us'], 'us  alone', (r.get('isolated') or {}).get('avg_launch_us'))"
done; done; done
