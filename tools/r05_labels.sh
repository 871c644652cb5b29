#!/bin/bash
# round 5: config 3's fused part (256^3 x 512 bf16 + label histogram) with the histogram counted per run of equal classes
# (SAF_WIN_LABEL_RUNS=1, the default) and per hit (=0), panoptic maps iid per pixel and consistent in 3-D, depth A and B
for k in A B; do for lk in iid world; do
VARIANTS="base LABEL_RUNS=0" BENCH_ARGS="--depth-kind $k --labels --feat-dtype bf16 --label-kind $lk --no-side --end-to-end 0 --no-pmc" bash tools/win_variants.sh gpurun_out/r05lab 2>&1 | grep "overlap=1" | sed "s/^/depth $k labels $lk: /"
done; done
