"""Post-process a rocprofv3 --kernel-trace CSV into what explains a block's wall time: per kernel family the time a dispatch waits
behind its predecessor IN THE SAME HARDWARE QUEUE (start - previous end: the CP could not start it: no block slot / a barrier), its
duration, and how many trace kernels are resident at once.  usage: r5_timeline.py <kernel_trace.csv> <out.json>"""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def fam(n):
    for k in ("trace_kernel", "classify_kernel", "upload_kernel", "deinterleave", "ncclDevKernel", "sky_kernel", "fillBuffer", "copyBuffer"):
        if k in n: return k
    return "other"
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), fam(r["Kernel_Name"]), r.get("Stream_Id", "")))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
byq = collections.defaultdict(list)
for e in ev: byq[e[2]].append(e)
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for q, L in byq.items():
    for a, b in zip(L, L[1:]):
        gaps[b[3]].append(max(0, b[0] - a[1]))
    for e in L: durs[e[3]].append(e[1] - e[0])
def st(v):
    v = sorted(v); n = len(v)
    return {"n": n, "mean_us": round(sum(v) / max(1, n) / 1e3, 1), "p50_us": round(v[n // 2] / 1e3, 1) if n else 0, "p90_us": round(v[int(n * 0.9)] / 1e3, 1) if n else 0, "max_us": round(v[-1] / 1e3, 1) if n else 0,
            "total_ms": round(sum(v) / 1e6, 2)}
# resident trace kernels over time
pts = []
for s, e, q, f, _ in ev:
    if f == "trace_kernel": pts += [(s, 1), (e, -1)]
pts.sort(); cur = 0; last = pts[0][0] if pts else 0; area = 0; peak = 0
for t, d in pts:
    area += cur * (t - last); last = t; cur += d; peak = max(peak, cur)
out = {"source": sys.argv[1].split("/")[-1], "dispatches": len(ev), "hardware_queues_used": len(byq), "streams": len(set(e[4] for e in ev)),
       "span_ms": round((t1 - t0) / 1e6, 2),
       "trace_kernels_resident": {"mean": round(area / max(1, (pts[-1][0] - pts[0][0])) if pts else 0, 2), "peak": peak},
       "duration": {k: st(v) for k, v in durs.items()},
       "wait_behind_predecessor_in_queue": {k: st(v) for k, v in gaps.items()},
       "note": "wait = start of a dispatch minus the end of the dispatch before it in the same hardware queue (>= 0): how long the command processor "
               "could not start it although its predecessor had finished - block slots taken by other queues' persistent kernels, barrier packets of cross-stream waits"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out)[:1500])
