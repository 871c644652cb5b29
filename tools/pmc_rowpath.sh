#!/bin/bash
# PMC passes over the row kernel's memory path (run on a GPU box; two counters per pass -- more of one block exceed what
# the hardware collects at once and rocprofv3 aborts): TA / TCP (L1) / TCC (L2) busy and stall counters of
# fuse_window_kernel for one 128-frame window (no classification beside it).  Usage: bash tools/pmc_rowpath.sh <outdir>
OUT=${1:-gpurun_out/pmcrow}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --cpu-frames 0 --frames 128 --steps 1 --warmup 0 --no-profile-events"
run() { name=$1; shift; timeout -k 5 80 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $B > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"; }
run a GRBM_GUI_ACTIVE TA_TA_BUSY_sum &&
run b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
run c TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum &&
run d TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum &&
run e TCC_BUSY_sum TCC_TAG_STALL_sum &&
run f TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
python3 - <<PY
import csv, glob, json, collections, os
out = {}
for d in "abcdef":
    fs = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % d), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        if "fuse_window" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c, v in acc.items():
        out[c] = v / n[c]
    ks = sorted(glob.glob("$OUT/%s/*/*_kernel_trace.csv" % d), key=os.path.getmtime)
    if ks:
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(ks[-1])) if "fuse_window" in r["Kernel_Name"]]
        if du:
            out["us_pass_" + d] = sum(du) / len(du)
json.dump(out, open("$OUT/rowpath_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
