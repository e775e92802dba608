#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/<workload>/ (written by profiles/collect.sh on the GPU box) into the summaries that are committed:
<tag>_kernel_stats[_<workload>].csv, <tag>_pmc[_<workload>].json, <tag>_bench_<workload>[_profiled].json and pmc_traffic.json (default
workload; read by bench.py for roofline.traffic and valu.issue).  Usage: summarize.py <tag> [destination directory]."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
top = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles")     # on the GPU box: a directory under gpurun_out/
os.makedirs(dst, exist_ok=True)


def find(src, sub, suffix):
    g = glob.glob(os.path.join(src, sub, "**", f"*{suffix}"), recursive=True)
    return g[0] if g else None


def per_kernel(src, sub):
    f = find(src, sub, "counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out[name]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


for workload in ("default", "euler", "mesh", "4k"):
    src = os.path.join(top, workload)
    if not os.path.isdir(src):
        continue
    sfx = "" if workload == "default" else "_" + workload
    stats = find(src, "stats", "kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats{sfx}.csv"))
    stats1 = find(src, "stats_f1", "kernel_stats.csv")
    if stats1:
        shutil.copy(stats1, os.path.join(dst, f"{tag}_kernel_stats_1frame_in_flight.csv"))
    for name, out in (("bench_line.json", f"{tag}_bench_{workload}.json"), ("bench_line_profiled.json", f"{tag}_bench_{workload}_profiled.json")):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 2:
            shutil.copy(p, os.path.join(dst, out))
    summary = {}
    for sub in ("fetch", "write", "sq", "sq2"):
        for k, cs in per_kernel(src, sub).items():
            if "bhray" not in k:
                continue
            d = summary.setdefault(k, {})
            for c, v in cs.items():
                if c == "_dur_ns":
                    d.setdefault("launches_" + sub, len(v))
                    d.setdefault("avg_dur_us_" + sub, sum(v) / len(v) / 1e3)
                else:
                    d[c + "_avg_per_launch"] = sum(v) / len(v)
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc{sfx}.json"), "w"), indent=1, sort_keys=True)
    # roofline.traffic for the dominant kernel of the workload: the trace kernel with the most launches
    cands = [k for k in summary if "trace_kernel" in k and "FETCH_SIZE_avg_per_launch" in summary[k] and "WRITE_SIZE_avg_per_launch" in summary[k]]
    cands.sort(key=lambda k: -summary[k].get("launches_fetch", 0))
    if cands:
        k = cands[0]
        fetch_kb, write_kb = summary[k]["FETCH_SIZE_avg_per_launch"], summary[k]["WRITE_SIZE_avg_per_launch"]
        traffic = {
            "kernel": k, "tag": tag, "workload": workload,
            "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
            # MI355X_MICROARCH.md (HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B under-counts 128 B requests by 2x on gfx950
            # (calibrated there for wide coalesced streams); WRITE_SIZE is uncalibrated.  Both forms are kept.
            "bytes_per_launch_raw": (fetch_kb + write_kb) * 1024.0,
            "bytes_per_launch_corrected": (2.0 * fetch_kb + write_kb) * 1024.0,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the same bench command, averaged over all launches of the kernel",
        }
        if "SQ_INSTS_VALU_avg_per_launch" in summary[k]:
            traffic["valu_insts_per_launch"] = summary[k]["SQ_INSTS_VALU_avg_per_launch"]
            if "SQ_THREAD_CYCLES_VALU_avg_per_launch" in summary[k] and "SQ_ACTIVE_INST_VALU_avg_per_launch" in summary[k]:
                traffic["valu_active_lane_fraction"] = summary[k]["SQ_THREAD_CYCLES_VALU_avg_per_launch"] / (64.0 * summary[k]["SQ_ACTIVE_INST_VALU_avg_per_launch"])
        json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json" if workload == "default" else f"pmc_traffic_{workload}.json"), "w"), indent=1)
        print(workload, json.dumps(traffic))
    if stats:
        print(open(stats).read()[:900])
