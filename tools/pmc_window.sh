#!/bin/bash
# PMC passes for the two kernels of the windowed path (run on a GPU box): instruction mix / VALU activity and
# L2 hit rate.  Separate passes (counter slots); summarised into profiles/<tag>/window_pmc.json.
# Usage: bash tools/pmc_window.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/pmcwin_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --cpu-frames 0 --frames 128 --steps 1 --warmup 0 --no-profile-events"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $B > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS &&
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES &&
run c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<PY
import csv, glob, json, collections, os
out = {}
for d in "abc":
    fs = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % d), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        kn = r["Kernel_Name"]
        k = "classify_bricks_kernel" if ("classify_window" in kn or "classify_bricks" in kn) else "fuse_window_kernel" if "fuse_window" in kn else None
        if k:
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for (k, c), v in acc.items():
        out.setdefault(k, {})[c] = v / n[(k, c)]
        out[k]["launches_" + d] = n[(k, c)]
for k, v in out.items():
    if "SQ_WAVE_CYCLES" in v and "SQ_ACTIVE_INST_VALU" in v:
        v["valu_active_frac_of_wave_cycles"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 4)
    if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v:
        v["wait_frac_of_wave_cycles"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4)
    if "TCC_HIT_sum" in v and "TCC_REQ_sum" in v:
        v["l2_hit_rate"] = round(v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
os.makedirs("$OUT", exist_ok=True)
json.dump(out, open("$OUT/window_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
