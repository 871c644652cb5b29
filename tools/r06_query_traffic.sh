#!/bin/bash
# round 6, on the GPU box: counter traffic of the wide scans (BASELINE config 5), FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
# passes over `bench.py --query --query-wide-only`, per launch of query_wide3_kernel and epilogue -> $OUT/query_traffic.json
# (committed as profiles/r06/query_traffic.json: the default bench line quotes it as roofline.traffic of the scans).
OUT=${1:-gpurun_out/r06qt}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --query --query-wide-only --steps 2 --warmup 1 --cpu-frames 0"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- $B > $OUT/$c.json 2> $OUT/$c.err || echo "FAILED $c"
done
python3 - <<PY
import csv, glob, json, collections, os
names = {"0": "config5_raw_scores", "1": "config5_heat_maps", "2": "config5_row_argmax", "3": "config5_query_max"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "query_wide3_kernel" in kn and r["Counter_Name"] == c:
                epi = kn.split("query_wide3_kernel<")[1].split(">")[0].split(",")[-1].strip()
                acc[names.get(epi, epi)][c].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    rd = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 1024 * 2
    wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024
    out[k] = {"read": int(rd), "write": int(wr), "hbm_bytes_per_launch": int(rd + wr)}
out["method"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over bench.py --query --query-wide-only; per launch of query_wide3_kernel; FETCH_SIZE x 2 (gfx950), KiB -> bytes"
json.dump(out, open("$OUT/query_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -size +1M -delete
