#!/bin/bash
# Counter passes over the row kernel of one 512-frame job (rocprofv3 serialises dispatches under --pmc: every kernel runs
# alone).  Usage: bash tools/pmc_rows.sh <outdir> [bench.py args, e.g. --depth-kind B]
OUT=$1; shift
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python3 bench.py --cpu-frames 0 --frames 512 --steps 1 --warmup 0 --no-profile-events --no-side --end-to-end 0 --no-pmc $@"
run() { name=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $P > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"; }
run a GRBM_GUI_ACTIVE TA_TA_BUSY_sum &&
run b SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS &&
run c SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC &&
run d SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS &&
run e SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL &&
run f TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum &&
run g TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum &&
run h SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 - <<PY
import csv, glob, json, collections, os
out = {}
for d in sorted(glob.glob("$OUT/?")):
    fs = sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs: continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        if "fuse_window" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c, v in acc.items(): out[c] = v / n[c]
    ks = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
    if ks:
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(ks[-1])) if "fuse_window" in r["Kernel_Name"]]
        if du: out["us_pass_" + os.path.basename(d)] = sum(du) / len(du)
json.dump(out, open("$OUT/rows_pmc.json", "w"), indent=1)
for k in sorted(out): print("%-40s %18.1f" % (k, out[k]))
PY
find $OUT -name "*.csv" -delete
