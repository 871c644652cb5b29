#!/bin/bash
# round 5, on the GPU box: the wide scan on v_mfma_f32_16x16x32 (SAF_WIDE_MFMA=16: query_wide3_kernel) against the 32x32x16 form
# (query_wide2_kernel), same box, config 5's scans (16.8 M rows x 1000 queries, fp16): ms per scan.
run() { SAF_WIDE_MFMA=$1 timeout -k 10 300 python3 bench.py --query --query-wide-only 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mfma=$1', ' | '.join('%s %.2f' % (c['case'][:28], c['ms']) for c in d['cases']))"; }
for r in ${ORDER:-32 16 32 16}; do run $r; done
