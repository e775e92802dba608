#!/bin/bash
# N = 1: frames per batch at long blocks (a launch that holds more rays than the device has lanes lets waves refill): throughput and
# the lane occupancy of the step loop (counter build)
mkdir -p gpurun_out/n1b
for f in 1 2 4 8; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sequence none --no-extra-legs --sustained-steps 400 --frames-per-batch $f > gpurun_out/n1b/fpb$f.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/n1b/fpb$f.json').read().strip().splitlines()[-1]); print('fpb $f', d['value'], d['sustained']['mrays_per_s'])"
  python - <<P
import bhusie_amd as B
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
rp = B.RayPass(cfg, device=0, counters=True, speculative_levels=2, frames_per_batch=$f, frames_in_flight=4)
rp.set_textures(*T.textures(small=False)); rp.set_uniforms(*T.uniforms(integration_method=1, max_iterations=2000))
for i in range(4 * $f): rp.render()
rp.sync()
c = rp.counters(); s = rp.scheduling_counters()
print("   fpb $f last frame: steps", c["steps"], "wave_steps", s["wave_steps"], "occupancy", s["step_lane_occupancy"])
P
done
