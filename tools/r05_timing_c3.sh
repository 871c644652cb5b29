#!/bin/bash
# round 5, on the GPU box: the row kernel's per-phase wave cycles (SAF_WIN_TIMING) for config 3's job (bf16 volume, panoptic
# labels) next to the fp32 headline, alone (SAF_WIN_OVERLAP=0) -- where a bf16 wave's time goes when its rows are half as long.
OUT=${1:-gpurun_out/r05t3}
mkdir -p $OUT
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v saf_window.o | tr "\n" " ")
/opt/rocm/bin/hipcc $FLAGS -DSAF_WIN_TIMING -c $C/saf_window.hip -o /tmp/win_timing.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_timing.so /tmp/win_timing.o $OTHERS || exit 1
B="python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --no-pmc --steps 1 --warmup 0"
for cfg in "" "--feat-dtype bf16" "--feat-dtype bf16 --labels" "--feat-dtype bf16 --labels --label-kind world"; do
  for ov in 0 1; do
    SAF_LIB_PATH=/tmp/libsaf_timing.so SAF_WIN_OVERLAP=$ov timeout -k 10 200 $B $cfg 2>&1 >/dev/null | grep "win timing" | tail -2 | sed "s/^/[${cfg:-f32}] overlap=$ov: /"
  done
done | tee $OUT/row_phases_config3.txt
# and the plain times of the same jobs on the shipped build
for cfg in "" "--feat-dtype bf16" "--feat-dtype bf16 --labels" "--feat-dtype bf16 --labels --label-kind world"; do
  timeout -k 10 200 $B --steps 6 --warmup 2 $cfg 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[${cfg:-f32}]', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows/window', r['avg_launch_us'], 'alone', (r.get('isolated') or {}).get('avg_launch_us'), 'classify', d['kernel_breakdown']['sweep_us'])"
done | tee -a $OUT/row_phases_config3.txt
