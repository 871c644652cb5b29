#!/bin/bash
# development A/B of a query_wide2_kernel compile-time option on one box: bash tools/w2_ab.sh "<-D flags>" ...
run() { timeout -k 10 300 python3 bench.py --query --query-wide-only 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', ' | '.join('%s %.2f' % (c['case'][:28], c['ms']) for c in d['cases']))"; }
run "default"
for f in "$@"; do
  (cd spatially_aware_ai_amd/csrc && touch saf_query_wide.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $f" > /dev/null 2>&1) && run "$f"
done
