#!/bin/bash
# development A/B of the wide scan's geometry (SAF_WIDE_ROWS) on one box, built with the flags given: bash tools/w2_env_ab.sh "<-D flags>" rows...
run() { SAF_WIDE_ROWS=$1 timeout -k 10 300 python3 bench.py --query --query-wide-only 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows=$1', ' | '.join('%s %.2f' % (c['case'][:28], c['ms']) for c in d['cases']))"; }
F="$1"; shift
(cd spatially_aware_ai_amd/csrc && touch saf_query_wide.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $F" > /dev/null 2>&1)
for r in "$@"; do run $r; done
(cd spatially_aware_ai_amd/csrc && touch saf_query_wide.hip && make > /dev/null 2>&1)
