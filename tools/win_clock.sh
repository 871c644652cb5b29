#!/bin/bash
# development: the shader clock while the row kernel runs, beside the classification and alone (SAF_WIN_TIMING build)
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -DSAF_WIN_TIMING" > /dev/null 2>&1)
for ov in 1 0; do echo "SAF_WIN_OVERLAP=$ov"; SAF_WIN_OVERLAP=$ov timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 3 --warmup 1 --no-profile-events 2>&1 | grep "shader clock" | tail -3; done
(cd spatially_aware_ai_amd/csrc && touch saf_window.hip && make > /dev/null 2>&1)
