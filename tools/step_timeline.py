#!/usr/bin/env python3
"""Timeline of ONE timed step of bench.py from a rocprofv3 --kernel-trace CSV: every kernel launch of the step with its start
(ms from the step's first kernel), duration and stream -- where the milliseconds outside the row kernels go.
Usage: python tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 1 = the last step]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("saf::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]
# a step of the headline job contains exactly 4 fuse_window_kernel launches (512 frames, 128-frame windows)
idx = [i for i, r in enumerate(rows) if "fuse_window_kernel" in r["Kernel_Name"]]
per = 4
end = len(idx) - (back - 1) * per
first_row, last_row = idx[end - per], idx[end - 1]
# the step starts at the first kernel after the previous step's last row kernel (+ its clear_unwritten)
prev_last = idx[end - per - 1] if end - per - 1 >= 0 else -1
start_i = prev_last + 1
while start_i < first_row and "clear_unwritten" in rows[start_i]["Kernel_Name"]:
    start_i += 1
stop_i = last_row
while stop_i + 1 < len(rows) and "clear_unwritten" in rows[stop_i + 1]["Kernel_Name"]:
    stop_i += 1
t0 = int(rows[start_i]["Start_Timestamp"])
print(f"{'start ms':>9s} {'dur ms':>8s}  {'queue':>5s}  kernel")
busy_end = t0
for r in rows[start_i:stop_i + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", r.get("Stream_Id", "?"))
    print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f}  {q:>5s}  {short(r['Kernel_Name'])}")
print("step (first kernel start -> last kernel end): %.3f ms" % ((int(rows[stop_i]["End_Timestamp"]) - t0) / 1e6))
