#!/bin/bash
# PMC passes over the window's kernels (run on a GPU box): the row kernel in the form SAF_WIN_FORM selects (sums / rows / bricks)
# and the classification, one after the other on one stream (SAF_WIN_OVERLAP=0): HBM bytes (FETCH_SIZE x 2 on gfx950,
# WRITE_SIZE), wave activity, LDS, L1 -> L2 requests, TA.  Default: one fresh 128-frame window; FRAMES=512: the whole job
# (per-launch averages over its four windows: the first is fresh, the others read most of their rows).
# Usage: [FRAMES=512] bash tools/pmc_brick.sh <outdir> [extra bench args]
OUT=${1:-gpurun_out/pmcbrick}; shift
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --cpu-frames 0 --frames ${FRAMES:-128} --steps 1 --warmup 0 --no-profile-events --no-side --end-to-end 0 --no-pmc $*"
export SAF_WIN_OVERLAP=0
run() { name=$1; shift; timeout -k 5 100 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $B > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"; }
run a FETCH_SIZE &&
run b WRITE_SIZE &&
run c SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS &&
run d SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run e TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum &&
run f GRBM_GUI_ACTIVE TA_TA_BUSY_sum &&
run g TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run h TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum
python3 - <<PY
import csv, glob, json, collections, os
def kind(name):
    if "fuse_window" in name:
        return "rows"
    if "classify_bricks" in name or "classify_window" in name:
        return "classify"
    if "fuse_brick" in name:  # fuse_brick_kernel<CPL, SUM, BF16, BUILD>: the last template argument tells the two kernels apart
        return "build" if name.split(">(")[0].rstrip().endswith("true") else "walk"
    return None
out = {}
for d in "abcdefgh":
    fs = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % d), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        k = kind(r["Kernel_Name"])
        if k:
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for (k, c), v in acc.items():
        out.setdefault(k, {})[c] = v / n[(k, c)]
    ks = sorted(glob.glob("$OUT/%s/*/*_kernel_trace.csv" % d), key=os.path.getmtime)
    if ks:
        du = collections.defaultdict(list)
        for r in csv.DictReader(open(ks[-1])):
            k = kind(r["Kernel_Name"])
            if k:
                du[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in du.items():
            out.setdefault(k, {})["us_pass_" + d] = sum(v) / len(v)
for k, o in out.items():
    if "FETCH_SIZE" in o:
        o["hbm_read_GB"] = o["FETCH_SIZE"] * 1024 * 2 / 1e9   # KiB, and the gfx950 halving
    if "WRITE_SIZE" in o:
        o["hbm_write_GB"] = o["WRITE_SIZE"] * 1024 / 1e9
    if "TCP_TCC_READ_REQ_sum" in o:
        o["l1_to_l2_read_GB_at_128B"] = o["TCP_TCC_READ_REQ_sum"] * 128 / 1e9
json.dump(out, open("$OUT/brick_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
