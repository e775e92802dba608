#!/bin/bash
# BHRAY_F_TEMPORAL: how much of a frame are the full-size grids of its (nearly empty) fix-up trace launches?  BHRAY_FIXUP_BLOCKS = grid of those launches
mkdir -p gpurun_out/fix
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0 --sequence orbit --steps 20 --warmup 5"
for fb in 0 256 64 16; do
  for mode in n1 r3; do
    extra=""; [ $mode = r3 ] && extra="--emulate-world 8 --emulate-rank 3"
    BHRAY_FIXUP_BLOCKS=$fb timeout 600 python bench.py $B $extra > gpurun_out/fix/${mode}_$fb.json 2>/dev/null
    python -c "
import json; d=json.loads(open('gpurun_out/fix/${mode}_$fb.json').read().strip().splitlines()[-1]); s=d['sequence']; print('fixup blocks $fb $mode', 'ladder', s['ladder']['ms_per_step'], s['ladder']['latency_ms_one_frame_in_flight'], 'temporal', s['temporal']['ms_per_step'], s['temporal']['latency_ms_one_frame_in_flight'], 'verified', s['verified_frames'])"
  done
done
