#!/bin/bash
# bash tools/cls_guard_verify.sh ["-D flags"]: the classification's self-check (SAF_CLS_VERIFY=1: both pixel paths, disagreements in
# stats[7]) over several image sizes -- tools/cls_guard_verify.py; with flags (e.g. -DSAF_CLS_GUARD_EPS=0.0f: the counter's own
# test) the library is rebuilt with them first and restored afterwards
if [ -n "$1" ]; then (cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC $1" > /dev/null 2>&1); fi
SAF_CLS_VERIFY=1 timeout -k 10 500 python3 tools/cls_guard_verify.py
if [ -n "$1" ]; then (cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make > /dev/null 2>&1); fi
