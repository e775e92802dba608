#!/bin/bash
# N-way row tiling emulated on ONE GPU, every rank's tiles in turn (before the gather): the slowest rank is what a real N-GPU frame costs.
# Writes gpurun_out/emulate_all_ranks.json (copied to profiles/r02_emulate_all_ranks.json).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT; mkdir -p gpurun_out
python - <<'PY'
import json, subprocess, sys
out = {}
for n in (2, 4, 8):
    for tag, extra in (("long_blocks", ["--steps", "192", "--warmup", "32"]), ("20_frame_blocks", ["--steps", "20", "--warmup", "5"])):
        ms = []
        for r in range(n):
            p = subprocess.run([sys.executable, "bench.py", "--emulate-world", str(n), "--emulate-rank", str(r), "--no-cpu-baseline", "--min-seconds", "0.4"] + extra, capture_output=True, text=True)
            ms.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["ms_per_step"])
        out[f"n{n}_{tag}"] = {"ms_per_frame_by_rank": ms, "slowest": max(ms), "fastest": min(ms)}
        print(n, tag, ms, flush=True)
base = {}
for tag, extra in (("long_blocks", ["--steps", "192", "--warmup", "32"]), ("20_frame_blocks", ["--steps", "20", "--warmup", "5"])):
    p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--min-seconds", "0.4"] + extra, capture_output=True, text=True)
    base[tag] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["ms_per_step"]
out["n1"] = base
for n in (2, 4, 8):
    for tag in ("long_blocks", "20_frame_blocks"):
        out[f"n{n}_{tag}"]["scaling_of_the_slowest_rank"] = round(base[tag] / out[f"n{n}_{tag}"]["slowest"], 3)
json.dump(out, open("gpurun_out/emulate_all_ranks.json", "w"), indent=1)
print(json.dumps({k: v.get("scaling_of_the_slowest_rank") for k, v in out.items() if k != "n1"}))
PY
