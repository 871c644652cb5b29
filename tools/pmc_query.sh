#!/bin/bash
# PMC passes over `bench.py --query --query-wide-only` (run on a GPU box): where do the wide scan's wave cycles go, and
# at what clock?  Separate passes (counter slots); summarised per kernel into $OUT/query_pmc.json.
# Usage: bash tools/pmc_query.sh <outdir>
OUT=${1:-gpurun_out/pmcq}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --query --query-wide-only --steps 2 --warmup 1 --cpu-frames 0"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $B > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES &&
run b GRBM_GUI_ACTIVE GRBM_COUNT &&
run c SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM &&
run d FETCH_SIZE &&   # HBM bytes (KiB; x 2 on gfx950 for FETCH_SIZE: MI355X_MICROARCH.md), separate passes
run e WRITE_SIZE
python3 - <<PY
import csv, glob, json, collections, os
out = collections.defaultdict(dict)
for d in "abcde":
    fs = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % d), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        kn = r["Kernel_Name"]
        if "query_wide3_kernel" not in kn:
            continue
        k = kn.split("query_wide3_kernel")[1].split("(")[0]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for (k, c), v in acc.items():
        out[k][c] = v / n[(k, c)]
    # kernel durations of this pass
    ks = sorted(glob.glob("$OUT/%s/*/*_kernel_trace.csv" % d), key=os.path.getmtime)
    if ks:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(ks[-1])):
            if "query_wide3_kernel" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("query_wide3_kernel")[1].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            out[k]["us_pass_" + d] = sum(v) / len(v)
json.dump(out, open("$OUT/query_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
PY
