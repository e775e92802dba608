#!/bin/bash
# steps per batch (BHRAY_REL_BATCH) and refill threshold (BHRAY_REFILL_MIN) again, now that the idle lanes are accounted for: N = 1 (20-frame
# blocks / 200-frame block) and rank 3 of an 8-way partition (batches of 10 frames: launches that refill)
mkdir -p gpurun_out/bk
B="--no-cpu-baseline --sequence none --no-extra-legs"
for round in 1 2; do
for v in default rb8 rb4 rb32 rm8 rb8rm8; do
  if [ $v = default ]; then L=$PWD/bhusie_amd/libbhray.so; else L=$PWD/profiles/variants/libbhray_$v.so; fi
  BHRAY_LIB=$L timeout 300 python bench.py $B --steps 20 --warmup 5 --sustained-steps 200 > gpurun_out/bk/n1_$v.json 2>/dev/null
  BHRAY_LIB=$L timeout 300 python bench.py $B --steps 20 --warmup 5 --sustained-steps 400 --emulate-world 8 --emulate-rank 3 > gpurun_out/bk/r3_$v.json 2>/dev/null
  python -c "
import json
a=json.loads(open('gpurun_out/bk/n1_$v.json').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/bk/r3_$v.json').read().strip().splitlines()[-1])
print('$v', 'N=1', a['value'], a['sustained']['mrays_per_s'], ' rank 3/8 ms per frame', b['ms_per_step'], b['sustained']['ms_per_step'])"
done
done
