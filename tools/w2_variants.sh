#!/bin/bash
# Development probe (run on a GPU box): builds variants of the wide-scan kernel with parts of a step removed
# (SAF_W2_NO_FEATLOAD: rows loaded once; SAF_W2_NO_TEXTLOAD: no transfer of the next text tile; SAF_W2_NO_BARRIER;
# SAF_W2_NO_FAST: epilogue in front of the MFMA block instead of between the MFMAs) and times the cases of
# `bench.py --query` with each, same box, same process order.  The variants' RESULTS are wrong; only timings count.  Usage: bash tools/w2_variants.sh <outdir>
OUT=${1:-gpurun_out/w2v}
mkdir -p $OUT
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v saf_query_wide.o | tr "\n" " ")
# VARIANTS: space-separated; a variant is `base` or switches joined by '+', e.g. NO_DMA+NO_STAGGER
for v in ${VARIANTS:-base NO_DMA NO_STAGGER NO_DMA+NO_STAGGER}; do
  tag=$(echo $v | tr '+' '_')
  def=""; [ "$v" != base ] && def=$(echo $v | sed 's/^/-DSAF_W2_/; s/+/ -DSAF_W2_/g')
  /opt/rocm/bin/hipcc $FLAGS $def -c $C/saf_query_wide.hip -o /tmp/qw_$tag.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_$tag.so /tmp/qw_$tag.o $OTHERS && \
  SAF_LIB_PATH=/tmp/libsaf_$tag.so python3 bench.py --query --query-wide-only --steps 3 --warmup 1 --cpu-frames 0 > $OUT/$tag.json 2> $OUT/$tag.err
  python3 - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1])
    print("$tag", " | ".join("%s %.2f ms" % (c["case"][:22], c["ms"]) for c in j["cases"]))
except Exception as e:
    print("$tag FAILED", e)
PY
done
