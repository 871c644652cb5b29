#!/bin/bash
# round 6, on the GPU box: the row kernel's fabric READ requests by size (TCC_EA0_RDREQ_128B / _64B / _32B) for the headline job and for
# config 3's: FETCH_SIZE is requests x 64 B whatever their size, so the guide's "x 2" is right for 128-byte requests only.
OUT=${1:-gpurun_out/r06c3}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for case in "f32_nolabels" "bf16_nolabels --feat-dtype bf16" "bf16_labels_iid --feat-dtype bf16 --labels" "f32_labels_iid --labels"; do
  set -- $case; name=$1; shift
  rm -rf $OUT/trace
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --cpu-frames 0 --steps 1 --warmup 0 --no-profile-events --no-side --end-to-end 0 --no-pmc --api-b1 0 "$@" > /dev/null 2> $OUT/err_${name}_reqsize.txt || { echo "$name FAILED"; tail -3 $OUT/err_${name}_reqsize.txt; continue; }
  python3 - <<PY
import csv, glob
per = {}
for f in glob.glob("$OUT/trace/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fuse_window_kernel" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
v = {k: sum(x) / 4 for k, x in per.items()}
print("$name", " ".join("%s=%.6g" % kv for kv in sorted(v.items())))
if "TCC_EA0_RDREQ_128B_sum" in v:
    b = 128 * v["TCC_EA0_RDREQ_128B_sum"] + 64 * v.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * v.get("TCC_EA0_RDREQ_32B_sum", 0)
    print("   read bytes per window by request size: %.3f GB   (FETCH_SIZE x 2 would say %.3f GB)" % (b / 1e9, v["TCC_EA0_RDREQ_sum"] * 128 / 1e9))
PY
done
rm -rf $OUT/trace
