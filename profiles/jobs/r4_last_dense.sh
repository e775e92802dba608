#!/bin/bash
# the LAST level's trace launch with the dense build (6 waves per SIMD) while the batch's coarse launches keep the latency build (thin dealing)
mkdir -p gpurun_out/ld
B="--no-cpu-baseline --no-extra-legs --sequence none --steps 20 --warmup 5 --sustained-steps 0"
for v in 0 1; do
  for t in "r0:--emulate-world 8 --emulate-rank 0" "r3:--emulate-world 8 --emulate-rank 3" "r5:--emulate-world 8 --emulate-rank 5" "q1:--emulate-world 4 --emulate-rank 1" "k4:--width 3840 --height 2160 --emulate-world 8 --emulate-rank 4" "h0:--emulate-world 2 --emulate-rank 0"; do
    n=${t%%:*}; a=${t#*:}
    BHRAY_LAST_LEVEL_DENSE=$v timeout 300 python bench.py $B $a > gpurun_out/ld/${n}_$v.json 2>/dev/null
  done
  python -c "
import json
g=lambda n: json.loads(open('gpurun_out/ld/%s_$v.json' % n).read().strip().splitlines()[-1])['ms_per_step']
print('last level dense $v:', {n: g(n) for n in ('r0','r3','r5','q1','k4','h0')})"
done
