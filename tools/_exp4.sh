timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_win.json 2> gpurun_out/bench_win.err || { tail -5 gpurun_out/bench_win.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_win.json").read().strip().splitlines()[-1])
print(d["value"], json.dumps(d["roofline"]), json.dumps(d["kernel_breakdown"]), json.dumps(d["cpu_baseline"]))
PY
