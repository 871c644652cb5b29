#!/bin/bash
# development: the 16x16x32 wide scan built with extra flags, one box: bash tools/w3_flags_ab.sh "<-D flags>" [mfma ...]
run() { SAF_WIDE_MFMA=$1 timeout -k 10 300 python3 bench.py --query --query-wide-only 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$F] mfma=$1', ' | '.join('%s %.2f' % (c['case'][:28], c['ms']) for c in d['cases']))"; }
F="$1"; shift
C=spatially_aware_ai_amd/csrc
OTHERS=$(ls $C/*.o | grep -v saf_query_wide.o | tr "\n" " ")
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function $F -c $C/saf_query_wide.hip -o /tmp/qw_f.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_qw.so /tmp/qw_f.o $OTHERS || exit 1
export SAF_LIB_PATH=/tmp/libsaf_qw.so
for r in "$@"; do run $r; done
