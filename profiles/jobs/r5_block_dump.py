"""Dump the dispatches of the LAST 60 ms of a rocprofv3 kernel trace as a compact time-ordered text (ms relative, queue, family, duration)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def fam(n):
    for k, s in (("trace_kernel", "TRACE"), ("classify_kernel", "cls"), ("upload_kernel", "upl"), ("deinterleave", "DEINT"), ("ncclDevKernel", "NCCL"), ("sky_kernel", "sky"), ("fillBuffer", "fill"), ("copyBuffer", "copy")):
        if k in n: return s
    return "other"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), fam(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Stream_Id", "")) for r in rows)
tend = max(e[1] for e in ev)
win = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
sel = [e for e in ev if e[0] >= tend - win * 1e6]
t0 = sel[0][0]
qmap = {}
for s, e, q, f, g, st in sel:
    qi = qmap.setdefault((q, st), len(qmap))
    print("%8.3f %8.3f q%02d %-5s grid %s" % ((s - t0) / 1e6, (e - s) / 1e6, qi, f, g))
