#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python - > gpurun_out/exp60.log 2>&1 <<'PY'
import json, subprocess, sys
for n in (8, 4):
    for stripe in (9, 18, 27, 54):
        for tag, extra in (("long", ["--steps", "192", "--warmup", "32"]), ("short", ["--steps", "20", "--warmup", "5"])):
            ms = []
            for r in range(n):
                p = subprocess.run([sys.executable, "bench.py", "--emulate-world", str(n), "--emulate-rank", str(r), "--stripe-rows", str(stripe), "--no-cpu-baseline", "--min-seconds", "0.3"] + extra, capture_output=True, text=True)
                ms.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["ms_per_step"])
            print("N", n, "stripe", stripe, tag, "slowest %.4f fastest %.4f mean %.4f" % (max(ms), min(ms), sum(ms) / len(ms)), flush=True)
PY
