#!/bin/bash
# bash tools/w2_stamps.sh ["extra -D flags"]: build with the step stamps, run tools/w2_stamps.py, rebuild the default library
(cd spatially_aware_ai_amd/csrc && touch saf_query_wide.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -DSAF_W2_STAMP $1" > /dev/null 2>&1) && timeout -k 10 300 python3 tools/w2_stamps.py
(cd spatially_aware_ai_amd/csrc && touch saf_query_wide.hip && make > /dev/null 2>&1)
