#!/bin/bash
# BHRAY_F_GATHER_SKY: tests, then 8 partitions on one GPU (RCCL send/recv to self) with the RGBA32F frame / the RGBA16F sky image gathered
mkdir -p gpurun_out/gsky
timeout 900 python -m pytest tests/test_gpu_multidevice.py -x -q -m gpu > gpurun_out/gsky/test.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|Error" gpurun_out/gsky/test.log | tail -3
B="--no-cpu-baseline --no-extra-legs --sequence none --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 5 --sustained-steps 200"
for res in "1080p:" "4k:--width 3840 --height 2160"; do
  name=${res%%:*}; a=${res#*:}
  for mode in frame sky; do
    extra=""; [ $mode = sky ] && extra="--gather-sky"
    timeout 600 python bench.py $B $a $extra > gpurun_out/gsky/${name}_$mode.json 2> gpurun_out/gsky/${name}_$mode.err; echo "$name $mode rc=$?"
    python - <<P
import json
d=json.loads(open("gpurun_out/gsky/${name}_$mode.json").read().strip().splitlines()[-1]); g=d["multi_gpu"] if "multi_gpu" in d else d.get("gather")
print("$name $mode", d["value"], d["ms_per_step"], (d.get("sustained") or {}).get("ms_per_step"), "verified", d.get("verified"), {k: g[k] for k in ("bytes_received_per_frame","receive_ms_per_batch","deinterleave_ms_per_batch","gathered_image") if g and k in g})
P
  done
done
