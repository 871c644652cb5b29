#!/bin/bash
# on the GPU box: the fp32 MFMA scan (query_mfma_kernel) with 256 and 512 threads per workgroup (SAF_Q_THREADS), bench.py --query's
# fp32 cases (L = 5 softmax, L = 63 surgery over the 256^3 x 512 fp32 volume): ms per scan.
for th in ${ORDER:-256 512 256 512}; do
  SAF_Q_THREADS=$th timeout -k 10 300 python3 bench.py --query 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('threads=$th', ' | '.join('%s %.2f' % (c['case'][:22], c['ms']) for c in d['cases'] if c['case'].startswith('L=')))"
done
