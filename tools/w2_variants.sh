#!/bin/bash
# Development probe (run on a GPU box): builds variants of the wide-scan kernel (SAF_W2_NO_DMA: text tiles staged
# through registers instead of LDS-DMA; SAF_W2_NO_STAGGER: all workgroups start in phase) and times the cases of
# `bench.py --query` with each, same box, same process order.  Usage: bash tools/w2_variants.sh <outdir>
OUT=${1:-gpurun_out/w2v}
mkdir -p $OUT
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS="$C/saf_fuse.o $C/saf_window.o $C/saf_query.o $C/saf_misc.o $C/saf_ccl.o $C/saf_mesh.o"
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
