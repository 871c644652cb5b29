#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/<tag>/: the rocprofv3 kernel
stats table, the bench line produced under the profiler, and the per-launch HBM traffic of the
fuse / sweep kernels from the PMC passes (FETCH_SIZE doubled: on gfx950 it reports half the bytes
of wide coalesced reads, guides/MI355X_MICROARCH.md §HBM; WRITE_SIZE is exact)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
dst = sys.argv[2] if len(sys.argv) > 2 else f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


ks = newest(f"{src}/trace/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks[0], f"{dst}/kernel_stats.csv")
    for r in csv.DictReader(open(ks[0])):
        if "saf::" in r["Name"]:
            short = r["Name"].split("saf::(anonymous namespace)::")[1].split("(")[0] if "saf::(anonymous namespace)::" in r["Name"] else r["Name"]
            short = short.split("<")[0] + ("<" + short.split("<", 1)[1] if "<" in short else "")
            print(f"{short:40s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
kh = newest(f"{src}/trace_headline/*/*_kernel_stats.csv")
if kh:
    shutil.copy(kh[0], f"{dst}/kernel_stats_headline.csv")
    if os.path.exists(f"{src}/bench_headline_under_rocprof.json"):
        shutil.copy(f"{src}/bench_headline_under_rocprof.json", f"{dst}/bench_headline_under_rocprof.json")
if os.path.exists(f"{src}/bench_under_rocprof.json"):
    shutil.copy(f"{src}/bench_under_rocprof.json", f"{dst}/bench_under_rocprof.json")


def name_of(k):
    if "classify_bricks_kernel" in k:
        return "classify_bricks_kernel"
    for n in ("fuse_window_kernel", "classify_window_kernel", "fuse_rows_kernel", "fuse_kernel", "sweep_kernel",
              "prep_rows_kernel", "prep_kernel", "query_kernel"):
        if n in k:
            return "fuse_kernel" if n in ("fuse_rows_kernel", "fuse_kernel") else n
    return None


out = {}
for d, cn in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    fs = newest(f"{src}/{d}/*/*_counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = name_of(r["Kernel_Name"])
        if k and r["Counter_Name"] == cn:
            agg[k].append(float(r["Counter_Value"]))
    out[cn] = {k: {"launches": len(v), "mean_KiB": sum(v) / len(v)} for k, v in agg.items()}
if out:
    json.dump(out, open(f"{dst}/pmc_fetch_write.json", "w"), indent=1)
    try:
        bj = json.loads(open(f"{src}/pmc_fetch.json").read().strip().splitlines()[-1])
        cfg = bj["config"]
        if "fuse_window_kernel" in out["FETCH_SIZE"]:  # the windowed path ran: per-window launches
            wt = {"grid": cfg["grid"], "dim": cfg["feat_dim"], "dtype": bj.get("dtype", "f32"), "frames_per_launch": int(round(cfg["frames_per_rank"] / out["FETCH_SIZE"]["fuse_window_kernel"]["launches"])),
                  "depth_kind": "B" if "depth-B" in cfg["workload"] else "A",
                  "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one 512-frame job; "
                            "FETCH_SIZE x2 (gfx950 correction), KiB -> bytes"}
            for kn, key in (("fuse_window_kernel", "hbm"), ("classify_bricks_kernel", "classify_hbm")):
                f = out["FETCH_SIZE"][kn]["mean_KiB"] * 1024 * 2
                w = out["WRITE_SIZE"][kn]["mean_KiB"] * 1024
                wt[f"{key}_read_bytes_per_launch"] = int(f)
                wt[f"{key}_write_bytes_per_launch"] = int(w)
                wt[f"{key}_bytes_per_launch"] = int(f + w)
                print("%s traffic per launch: read %.1f MB write %.1f MB" % (kn, f / 1e6, w / 1e6))
            json.dump(wt, open(f"{dst}/window_traffic.json", "w"), indent=1)
            raise SystemExit(0)
        f = out["FETCH_SIZE"]["fuse_kernel"]["mean_KiB"] * 1024 * 2
        w = out["WRITE_SIZE"]["fuse_kernel"]["mean_KiB"] * 1024
        t = {"grid": cfg["grid"], "dim": cfg["feat_dim"], "hbm_read_bytes_per_launch": int(f),
             "hbm_write_bytes_per_launch": int(w), "hbm_bytes_per_launch": int(f + w),
             "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, 24 launches each; "
                       "FETCH_SIZE x2 (gfx950 correction), KiB -> bytes"}
        json.dump(t, open(f"{dst}/fuse_traffic.json", "w"), indent=1)
        print("fuse traffic per launch: read %.1f MB write %.1f MB" % (f / 1e6, w / 1e6))
    except Exception as e:  # noqa: BLE001
        print("no traffic summary:", e)
