#!/bin/bash
# Development probe (run on a GPU box): builds variants of saf_window.hip (compile-time knobs without the SAF_WIN_ prefix,
# e.g. EMU=2: two copies of the map images with the pieces spread over them = the tap footprint of a window twice as long)
# and times the default bench.py with each, for the unit orders in XS (SAF_WIN_XCD=0 linear / 1 XCD-compact) and the
# classification overlap settings in OVS.  Same box, same process order.
# Usage: VARIANTS="base EMU=2" XS="0 1" OVS="0 1" bash tools/xcd_variants.sh <outdir>
OUT=${1:-gpurun_out/xcdv}
mkdir -p $OUT
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v saf_window.o | tr "\n" " ")
for v in ${VARIANTS:-base EMU=2}; do
  tag=$(echo $v | tr '+=' '__')
  def=""; [ "$v" != base ] && def=$(echo $v | sed 's/^/-DSAF_WIN_/; s/+/ -DSAF_WIN_/g')
  /opt/rocm/bin/hipcc $FLAGS $def -c $C/saf_window.hip -o /tmp/win_$tag.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_$tag.so /tmp/win_$tag.o $OTHERS || { echo "$tag BUILD FAILED"; continue; }
  for x in ${XS:-0 1}; do for ov in ${OVS:-0 1}; do
    SAF_WIN_XCD=$x SAF_WIN_OVERLAP=$ov SAF_LIB_PATH=/tmp/libsaf_$tag.so python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 ${BENCH_ARGS} > $OUT/${tag}_x${x}_ov$ov.json 2> $OUT/${tag}_x${x}_ov$ov.err
    python3 - <<PY
import json
try:
    j = json.loads(open("$OUT/${tag}_x${x}_ov$ov.json").read().strip().splitlines()[-1])
    print("$tag xcd=$x overlap=$ov", j["value"], "frames/s", j["ms_per_step"], "ms  classify", j["kernel_breakdown"]["sweep_us"], "us  rows", j["kernel_breakdown"]["fuse_us"], "us")
except Exception as e:
    print("$tag xcd=$x overlap=$ov FAILED", e)
PY
  done; done
done
