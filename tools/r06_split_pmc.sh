#!/bin/bash
# round 6, on the GPU box: instruction and wait counters of the split scan (query_split_kernel, L = 5 / L = 63 over 2^23 x 512 fp32 rows),
# one rocprofv3 --pmc pass per group of counters over tools/probe_qmfma.py -> $OUT/split_scan_pmc.json
OUT=${1:-gpurun_out/r06sp}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_CVT"
G3="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
G4="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"
i=0
for g in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/g$i -- python3 tools/probe_qmfma.py 8388608 5 63 > $OUT/g$i.txt 2> $OUT/g$i.err || echo "FAILED group $i"
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "query_split_kernel" in kn:
            t = [x.strip() for x in kn.split("query_split_kernel<")[1].split(">")[0].split(",")]
            acc["L5_softmax_last" if t[2] == "1" else "L63_surgery"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} for k, d in acc.items()}
out["method"] = "rocprofv3 --pmc, four passes over tools/probe_qmfma.py 8388608 5 63 (2^23 rows x 512 fp32); per launch of query_split_kernel, averages over the launches of a pass"
json.dump(out, open("$OUT/split_scan_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -size +1M -delete
