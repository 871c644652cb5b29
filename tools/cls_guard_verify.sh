#!/bin/bash
# bash tools/cls_guard_verify.sh: build with both pixel paths and the disagreement counter, run tools/cls_guard_verify.py, rebuild
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -DSAF_CLS_GUARD=2 $1" > /dev/null 2>&1) && timeout -k 10 500 python3 tools/cls_guard_verify.py
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make > /dev/null 2>&1)
