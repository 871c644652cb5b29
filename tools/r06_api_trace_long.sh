#!/bin/bash
# round 6, on the GPU box: the one-frame-per-call loop in its steady state -- a job of N (default 2048) calls under
# rocprofv3 --kernel-trace: per window the row kernel's start / end, the wait before it, and the classification launches beside it.
OUT=${1:-gpurun_out/r06apil}; N=${2:-2048}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --cpu-frames 0 --no-pmc --no-side --end-to-end 0 --steps 2 --warmup 1 --api-b1 $N > $OUT/api.json 2> $OUT/api.err
python3 - <<PY
import csv, glob, json
f = sorted(glob.glob("$OUT/trace/*/*_kernel_trace.csv"))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
st = [i for i, r in enumerate(rows) if "stage_frame_kernel" in r[2]]
api0 = st[-$N]
t0 = rows[api0][0]
rk = [(s, e) for s, e, k in rows[api0:] if "fuse_window" in k]
cl = [(s, e) for s, e, k in rows[api0:] if "classify_bricks" in k]
sg = [(s, e) for s, e, k in rows[api0:] if "stage_frame_kernel" in k]
prev_end = None
for w, (s, e) in enumerate(rk):
    beside = [(cs, ce) for cs, ce in cl if cs < e and ce > s]
    last_cls_before = max([ce for cs, ce in cl if ce <= s + 1000] or [0])
    nst = sum(1 for a, b in sg if a < e and b > s)
    print("rows(%d) %.2f .. %.2f (%.2f)  gap before %.2f  last classification before it ended %.2f ms earlier | beside: %s | %d staging kernels beside" % (
        w, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6 if prev_end else (s - t0) / 1e6,
        (s - last_cls_before) / 1e6, " ".join("%.2f" % ((ce - cs) / 1e6) for cs, ce in beside), nst))
    prev_end = e
d = json.loads(open("$OUT/api.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("api_b1"))
PY
find $OUT -name "*.csv" -size +1M -delete
