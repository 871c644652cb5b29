#!/bin/bash
# Development probe (run on a GPU box): builds variants of saf_window.hip with other compile-time knobs and times the
# default bench.py with each, overlap of the classification on and off, same box, same process order.
# Usage: VARIANTS="base P2=3 P2=2+SR2=6" bash tools/win_variants.sh <outdir>     (knob names without the SAF_WIN_ prefix)
OUT=${1:-gpurun_out/winv}
mkdir -p $OUT
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v saf_window.o | tr "\n" " ")
for v in ${VARIANTS:-base P2=3 P2=2}; do
  tag=$(echo $v | tr '+=' '__')
  def=""; [ "$v" != base ] && def=$(echo $v | sed 's/^/-DSAF_WIN_/; s/+/ -DSAF_WIN_/g')
  /opt/rocm/bin/hipcc $FLAGS $def -c $C/saf_window.hip -o /tmp/win_$tag.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_$tag.so /tmp/win_$tag.o $OTHERS || { echo "$tag BUILD FAILED"; continue; }
  for ov in 1 0; do
    SAF_WIN_OVERLAP=$ov SAF_LIB_PATH=/tmp/libsaf_$tag.so python3 bench.py --steps 5 --warmup 2 --cpu-frames 0 ${BENCH_ARGS} > $OUT/${tag}_ov$ov.json 2> $OUT/${tag}_ov$ov.err
    python3 - <<PY
import json
try:
    j = json.loads(open("$OUT/${tag}_ov$ov.json").read().strip().splitlines()[-1])
    print("$tag overlap=$ov", j["value"], "frames/s", j["ms_per_step"], "ms  classify", j["kernel_breakdown"]["sweep_us"], "us  rows", j["kernel_breakdown"]["fuse_us"], "us")
except Exception as e:
    print("$tag overlap=$ov FAILED", e)
PY
  done
done
