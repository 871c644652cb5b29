#!/bin/bash
# round 5, on the GPU box: the row kernel's per-phase wave cycles (SAF_WIN_TIMING) with the classification beside it
# (SAF_WIN_OVERLAP=1) and without (=0): which phases of a wave stretch when the two kernels share the CUs.
OUT=${1:-gpurun_out/r05t}
mkdir -p $OUT
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v saf_window.o | tr "\n" " ")
/opt/rocm/bin/hipcc $FLAGS -DSAF_WIN_TIMING -c $C/saf_window.hip -o /tmp/win_timing.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_timing.so /tmp/win_timing.o $OTHERS || exit 1
for k in A B; do for ov in 0 1; do
  SAF_LIB_PATH=/tmp/libsaf_timing.so SAF_WIN_OVERLAP=$ov timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --no-pmc --steps 1 --warmup 0 --depth-kind $k 2>&1 >/dev/null | grep "win timing" | tail -2 | sed "s/^/depth $k overlap=$ov: /"
done; done | tee $OUT/row_phases.txt
