#!/bin/bash
# round 5: the order-free kernel's tap ring and dense-group path, on and off, scenes B and A, same box
OUT=${1:-gpurun_out/r05ring}; mkdir -p $OUT
for k in ${KINDS:-B A}; do
VARIANTS="${VARIANTS:-base OF_RING=0 OF_DENSE=0}" BENCH_ARGS="--depth-kind $k --no-side --end-to-end 0 --no-pmc" bash tools/win_variants.sh $OUT 2>&1 | grep -v "^$" | sed "s/^/depth $k: /"
done
