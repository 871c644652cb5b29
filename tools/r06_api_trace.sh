#!/bin/bash
# round 6, on the GPU box: kernel timeline of one BULK step and of the one-frame-per-call API loop (bench.py --api-b1 512) from the same
# rocprofv3 --kernel-trace run: classification / row kernels / staging, ms from the first kernel of each.
OUT=${1:-gpurun_out/r06api}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --cpu-frames 0 --no-pmc --no-side --end-to-end 0 --steps 2 --warmup 1 --api-b1 512 > $OUT/api.json 2> $OUT/api.err
python3 - <<PY
import csv, glob, json
f = sorted(glob.glob("$OUT/trace/*/*_kernel_trace.csv"))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
def name_of(k):
    return "stage" if "stage_frame" in k else "rows" if "fuse_window" in k else "classify" if "classify_bricks" in k else "clear" if "clear_unwritten" in k else "prep" if "prep_rows" in k else "tiles" if "depth_" in k else None
def show(lo, hi, title):
    print("----", title)
    t0 = rows[lo][0]
    out = []; run = None; tiles = None
    for i in range(lo, hi):
        s, e, k = rows[i]
        n = name_of(k)
        if n is None: continue
        s, e = (s - t0) / 1e6, (e - t0) / 1e6
        if n == "stage":
            if run is None: run = [s, e, 1]
            else: run[1] = e; run[2] += 1
            continue
        if n == "tiles":
            if tiles is None: tiles = [s, e, 1]
            else: tiles[1] = e; tiles[2] += 1
            continue
        if run: out.append("stage x%d  %.2f .. %.2f ms" % (run[2], run[0], run[1])); run = None
        if tiles: out.append("tiles x%d  %.2f .. %.2f ms" % (tiles[2], tiles[0], tiles[1])); tiles = None
        out.append("%-8s %.2f .. %.2f ms  (%.2f)" % (n, s, e, e - s))
    if run: out.append("stage x%d  %.2f .. %.2f ms" % (run[2], run[0], run[1]))
    print("\n".join(out[:200]))
st = [i for i, r in enumerate(rows) if "stage_frame_kernel" in r[2]]
api0 = st[-512]
# the bulk step before the api loop: the last 4 row kernels ahead of the first stage launch of the warm-up api loop
first_stage = st[0]
rk = [i for i, r in enumerate(rows[:first_stage]) if "fuse_window" in r[2]]
lo = rk[-4]
while lo > 0 and name_of(rows[lo - 1][2]) in ("classify", "tiles", "prep") and rows[lo - 1][0] > rows[rk[-5]][1]: lo -= 1
show(lo, rk[-1] + 1, "bulk step (saf_fuse_frames, 512 frames in one call)")
show(api0, len(rows), "api loop (one frame per integrate_features call)")
d = json.loads(open("$OUT/api.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("api_b1"))
PY
find $OUT -name "*.csv" -size +1M -delete
