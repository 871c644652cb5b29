#!/usr/bin/env python3
"""Condenses tools/r05_cores.sh's counter passes: per kernel (classification / row kernel) and per schedule
(SAF_WIN_OVERLAP=1 -- the classification of window w + 1 queued beside window w's row kernel -- and =0), the mean counter
values per dispatch, the mean dispatch duration, and -- from the kernel trace of the same pass -- how much of the
classification's time overlapped a row kernel in time at all (a profiler that serialises dispatches to attribute counters
shows 0 here, and the "beside" column is then not a measurement of co-residency).  Writes <outdir>/coresident_pmc.json."""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]


def kind(name):
    if "classify_bricks" in name:
        return "classify"
    if "fuse_window" in name:
        return "rows"
    return None


res = {}
for ov in (1, 0):
    sched = "beside" if ov else "alone"
    for d in sorted(glob.glob(f"{out_dir}/pmc_ov{ov}/?")):
        cc = sorted(glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
        kt = sorted(glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
        if not cc:
            continue
        acc = collections.defaultdict(float)
        n = collections.Counter()
        for r in csv.DictReader(open(cc[-1])):
            k = kind(r["Kernel_Name"])
            if k:
                acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
                n[(k, r["Counter_Name"])] += 1
        for (k, c), v in acc.items():
            res.setdefault(k, {}).setdefault(c, {})[sched] = v / n[(k, c)]
        if kt:
            iv = collections.defaultdict(list)
            for r in csv.DictReader(open(kt[-1])):
                k = kind(r["Kernel_Name"])
                if k:
                    iv[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            p = os.path.basename(d)
            for k, v in iv.items():
                res.setdefault(k, {}).setdefault("us_pass_" + p, {})[sched] = sum(e - s for s, e in v) / len(v) / 1e3
            # share of the classification's time that lies inside some row kernel's interval
            tot = ovl = 0
            for s, e in iv.get("classify", []):
                tot += e - s
                for s2, e2 in iv.get("rows", []):
                    ovl += max(0, min(e, e2) - max(s, s2))
            if tot:
                res.setdefault("classify", {}).setdefault("time_inside_a_row_kernel_pass_" + p, {})[sched] = round(ovl / tot, 3)
for k, v in res.items():
    for c, d in v.items():
        if "beside" in d and "alone" in d and d["alone"]:
            d["beside_over_alone"] = round(d["beside"] / d["alone"], 3)
json.dump(res, open(f"{out_dir}/coresident_pmc.json", "w"), indent=1)
for k, v in res.items():
    print("==", k)
    for c, d in sorted(v.items()):
        print("  %-44s alone %16.1f  beside %16.1f  x %s" % (c, d.get("alone", float("nan")), d.get("beside", float("nan")), d.get("beside_over_alone")))
