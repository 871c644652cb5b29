#!/bin/bash
# round 6, on the GPU box: where config 3's HBM traffic goes (counter traffic 1.23 x algorithmic).  One 512-frame job
# per (case, counter pass) under rocprofv3 --pmc, the row kernel's launches summed per window.  Cases: bf16 rows without / with the
# ClipSeem side (bilinear rgb + label histogram), f32 rows with it.  Counters: FETCH_SIZE / WRITE_SIZE (KiB; the guide doubles
# FETCH_SIZE for wide coalesced reads) and the request counters behind them, which tell 32- / 64-byte requests from whole lines.
OUT=${1:-gpurun_out/r06c3}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*ATOMIC[A-Z0-9_]*" | sort -u > $OUT/tcc_counters.txt
for case in "bf16_nolabels --feat-dtype bf16" "bf16_labels_iid --feat-dtype bf16 --labels" "f32_labels_iid --labels" "bf16_labels_world --feat-dtype bf16 --labels --label-kind world"; do
  set -- $case; name=$1; shift
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_DRAM_sum"; do
    tag=$(echo $pass | tr ' ' '+')
    rm -rf $OUT/trace
    timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --cpu-frames 0 --steps 1 --warmup 0 --no-profile-events --no-side --end-to-end 0 --no-pmc --api-b1 0 "$@" > /dev/null 2> $OUT/err_${name}_$tag.txt || { echo "$name $tag FAILED"; continue; }
    python3 - <<PY
import csv, glob
per = {}
for f in glob.glob("$OUT/trace/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fuse_window_kernel" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("$name", " ".join("%s=%.6g/window(%d launches)" % (k, sum(v) / 4, len(v)) for k, v in sorted(per.items())))
PY
  done
done
rm -rf $OUT/trace
