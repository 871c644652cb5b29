#!/bin/bash
# dump the gfx950 assembly of ONE kernel of a .hip file: bash tools/kasm.sh <file.hip> <mangled-name prefix> [out.s] [extra flags]
cd "$(dirname "$0")/../spatially_aware_ai_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -S --cuda-device-only "$1" -o /tmp/kasm_all.s $4 2>&1 | grep -E "error" -A5
a=$(grep -n "^$2" /tmp/kasm_all.s | head -1 | cut -d: -f1)
awk -v a=$a 'NR>=a{print} NR>a&&/s_endpgm/{exit}' /tmp/kasm_all.s > "${3:-/tmp/k.s}"
wc -l "${3:-/tmp/k.s}"
