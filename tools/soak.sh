#!/bin/bash
# Runs bench.py over the shapes the windowed and the per-frame path must both survive (each run checks the
# fused frame count and the weight sum against the kernel counters).  Usage on a GPU box: bash tools/soak.sh
run() { echo "## $*"; timeout -k 10 300 python bench.py --cpu-frames 0 "$@" 2>&1 | grep -o "\"value\": [0-9.]*\|Error.*\|Assertion.*" | tr '\n' ' '; echo; }
run --steps 10 --warmup 1
run --steps 4 --warmup 1 --labels
run --steps 4 --warmup 1 --labels --depth-kind B
run --steps 4 --warmup 1 --dim 256
run --steps 2 --warmup 1 --dim 768 --frames 200
run --steps 2 --warmup 1 --dim 1024 --frames 100
run --steps 4 --warmup 1 --grid 128
run --steps 4 --warmup 1 --frames 70
run --steps 4 --warmup 1 --feat-dtype bf16 --labels
