mkdir -p gpurun_out/pmcw3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SAF_WIN_SPLIT=1 SAF_WIN_SERIAL=1
B="python3 bench.py --cpu-frames 0 --frames 64 --steps 1 --warmup 0 --no-profile-events"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmcw3/$name -- $B > gpurun_out/pmcw3/$name.json 2> gpurun_out/pmcw3/$name.err || echo "FAILED $name"; }
run f FETCH_SIZE && run h TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<'PY'
import csv,glob,collections
for d in "fh":
    fs=glob.glob(f"gpurun_out/pmcw3/{d}/**/*counter_collection.csv",recursive=True)
    if not fs: print(d,"no csv"); continue
    acc=collections.defaultdict(float); n=collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        kn=row["Kernel_Name"]
        key=("classify" if "classify" in kn else "window" if "fuse_window" in kn else None)
        if key:
            acc[(key,row["Counter_Name"])]+=float(row["Counter_Value"]); n[(key,row["Counter_Name"])]+=1
    for k in acc: print(d,k,acc[k]/max(1,n[k]),"per launch over",n[k])
PY
