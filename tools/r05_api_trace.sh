#!/bin/bash
# round 5, on the GPU box: kernel timeline of the one-frame-per-call API loop (bench.py --api-b1 512): where the staging kernels of
# the later frames run relative to the first flush's kernels.
OUT=${1:-gpurun_out/r05api}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --cpu-frames 0 --no-pmc --no-side --end-to-end 0 --steps 2 --warmup 1 --api-b1 512 > $OUT/api.json 2> $OUT/api.err
python3 - <<PY
import csv, glob, json
f = sorted(glob.glob("$OUT/trace/*/*_kernel_trace.csv"))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the LAST run of 512 stage_frame launches = the timed api loop
st = [i for i, r in enumerate(rows) if "stage_frame_kernel" in r[2]]
last = st[-512:]
t0 = rows[last[0]][0]
ev = []
for i in range(last[0], len(rows)):
    s, e, k = rows[i]
    name = "stage" if "stage_frame" in k else "rows" if "fuse_window" in k else "classify" if "classify_bricks" in k else "clear" if "clear_unwritten" in k else "prep" if "prep_rows" in k else "tiles" if "depth_" in k else None
    if name: ev.append((name, (s - t0) / 1e6, (e - t0) / 1e6))
# compress consecutive stage launches
out = []; run = None
for name, s, e in ev:
    if name == "stage":
        if run is None: run = [s, e, 1]
        else: run[1] = e; run[2] += 1
    else:
        if run: out.append("stage x%d  %.2f .. %.2f ms" % (run[2], run[0], run[1])); run = None
        out.append("%-8s %.2f .. %.2f ms" % (name, s, e))
if run: out.append("stage x%d  %.2f .. %.2f ms" % (run[2], run[0], run[1]))
print("\n".join(out[:120]))
d = json.loads(open("$OUT/api.json").read().strip().splitlines()[-1])
print(d.get("api_b1"))
PY
find $OUT -name "*.csv" -size +1M -delete
