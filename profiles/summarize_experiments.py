#!/usr/bin/env python3
"""gpurun_out/prof_exp (profiles/collect_experiments.sh) -> profiles/r02_experiments.json: per variant, the trace kernels' average
launch duration (rocprofv3 --kernel-trace --stats), FETCH_SIZE per launch, VALU instructions and active-lane fraction (PMC)."""
import collections
import csv
import glob
import json
import os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_exp")
out = {"bench_lines": open(os.path.join(src, "bench_lines.txt")).read().splitlines()}
for d in sorted(glob.glob(os.path.join(src, "*_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    e = out.setdefault(name, {})
    st = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        for r in csv.DictReader(open(st[0])):
            if "trace_kernel" in r["Name"] and int(r["Calls"]) > 50:
                e.setdefault("kernel_stats", []).append({"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                          "pct_of_kernel_time": float(r["Percentage"])})
    for sub in ("fetch", "sq"):
        cc = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
        if not cc:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(cc[0])):
            k = r["Kernel_Name"].split("(")[0]
            if "trace_kernel" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            n = max(len(v) for v in cs.values())
            if n < 50:
                continue
            row = {"kernel": k, "launches": n}
            for c, v in cs.items():
                row[c + "_per_launch"] = sum(v) / len(v)
            if "SQ_THREAD_CYCLES_VALU" in cs and "SQ_ACTIVE_INST_VALU" in cs:
                row["valu_active_lane_fraction"] = sum(cs["SQ_THREAD_CYCLES_VALU"]) / (64.0 * sum(cs["SQ_ACTIVE_INST_VALU"]))
            e.setdefault(sub, []).append(row)
json.dump(out, open(os.path.join(root, "profiles", "r02_experiments.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
