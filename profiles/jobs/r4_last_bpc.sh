#!/bin/bash
# persistent blocks per CU of the LAST level's trace launch only (the coarse launches keep the ctx's rule): one GPU and ranks of an 8-way partition
mkdir -p gpurun_out/lb
B="--no-cpu-baseline --no-extra-legs --sequence none --steps 20 --warmup 5 --sustained-steps 200"
for v in 0 3 4 6; do
  BHRAY_LAST_LEVEL_BPC=$v timeout 300 python bench.py $B > gpurun_out/lb/n1_$v.json 2>/dev/null
  BHRAY_LAST_LEVEL_BPC=$v timeout 300 python bench.py $B --emulate-world 8 --emulate-rank 3 > gpurun_out/lb/r3_$v.json 2>/dev/null
  BHRAY_LAST_LEVEL_BPC=$v timeout 300 python bench.py $B --emulate-world 8 --emulate-rank 0 > gpurun_out/lb/r0_$v.json 2>/dev/null
  BHRAY_LAST_LEVEL_BPC=$v timeout 300 python bench.py $B --width 3840 --height 2160 --emulate-world 8 --emulate-rank 4 > gpurun_out/lb/k4_$v.json 2>/dev/null
  python -c "
import json
g=lambda n: json.loads(open('gpurun_out/lb/%s_$v.json' % n).read().strip().splitlines()[-1])
a,b,c,d=g('n1'),g('r3'),g('r0'),g('k4')
print('last-level bpc $v: N=1', a['value'], a['sustained']['mrays_per_s'], '| rank 3/8', b['ms_per_step'], b['sustained']['ms_per_step'], '| rank 0/8', c['ms_per_step'], c['sustained']['ms_per_step'], '| 4K rank 4/8', d['ms_per_step'], d['sustained']['ms_per_step'])"
done
