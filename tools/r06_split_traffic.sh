#!/bin/bash
# round 6, on the GPU box: counter traffic of the reference's own scans over the 256^3 x 512 fp32 volume (query_split_kernel: L = 5
# softmax to the last column, L = 63 surgery), FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes over
# tools/probe_qmfma.py -> $OUT/split_scan_traffic.json (committed under profiles/r06/: `bench.py --query` quotes it as the scans'
# roofline.traffic).
OUT=${1:-gpurun_out/r06st}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 tools/probe_qmfma.py 16777216 5 63 > $OUT/$c.txt 2> $OUT/$c.err || echo "FAILED $c"
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "query_split_kernel" in kn and r["Counter_Name"] == c:
                t = [x.strip() for x in kn.split("query_split_kernel<")[1].split(">")[0].split(",")]
                acc["L5_softmax_last" if t[2] == "1" else "L63_surgery"][c].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    rd = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 1024 * 2
    wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024
    out[k] = {"read": int(rd), "write": int(wr), "hbm_bytes_per_launch": int(rd + wr), "launches": len(v["FETCH_SIZE"])}
out["method"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over tools/probe_qmfma.py 16777216 5 63; per launch of query_split_kernel; FETCH_SIZE x 2 (gfx950), KiB -> bytes"
json.dump(out, open("$OUT/split_scan_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -size +1M -delete
