#!/bin/bash
# BHRAY_F_TEMPORAL predicts a frame from the frame that held its slot position before: with S slots x B frames per batch that frame is S x B frames old.
# The orbit sequence (camera strafing 0.0066 rad per frame) per rank with shallower rings.
mkdir -p gpurun_out/tr
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0 --steps 20 --warmup 5 --sequence orbit --partition-feedback-temporal"
for cfg in "22:-1" "4:10" "2:10" "4:5" "8:2" "3:10"; do
  fif=${cfg%%:*}; fpb=${cfg#*:}
  for r in 3 7; do
    timeout 600 python bench.py $B --frames-in-flight $fif --frames-per-batch $fpb --emulate-world 8 --emulate-rank $r > gpurun_out/tr/r${r}_${fif}_$fpb.json 2>/dev/null
    python -c "
import json; d=json.loads(open('gpurun_out/tr/r${r}_${fif}_$fpb.json').read().strip().splitlines()[-1]); s=d['sequence']; print('slots $fif batch $fpb rank $r: ladder', s['ladder']['ms_per_step'], 'temporal', s['temporal']['ms_per_step'], 'fpb', s['frames_per_batch'])"
  done
done
