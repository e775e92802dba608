#!/bin/bash
mkdir -p gpurun_out/fc
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -2
B="--no-cpu-baseline --no-extra-legs --sequence none --steps 20 --warmup 5 --sustained-steps 400"
timeout 300 python bench.py $B --emulate-world 8 --emulate-rank 3 > gpurun_out/fc/r3.json 2>gpurun_out/fc/r3.err; echo "rank3 rc=$?"
timeout 300 python bench.py $B --gpus 4 --devices 0,0,0,0 > gpurun_out/fc/g4.json 2>gpurun_out/fc/g4.err; echo "gpus4 rc=$?"
timeout 300 python bench.py $B > gpurun_out/fc/n1.json 2>gpurun_out/fc/n1.err; echo "n1 rc=$?"
python - <<'P'
import json
for n in ("r3","g4","n1"):
    d=json.loads(open(f"gpurun_out/fc/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], "fpb", d["config"]["frames_per_batch"], "sustained", d["sustained"]["ms_per_step"], d["sustained"].get("frames_per_batch"), "verified", d["config"].get("verified_frames"))
P
tail -2 gpurun_out/fc/*.err
