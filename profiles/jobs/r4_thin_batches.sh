#!/bin/bash
# latency build: a batch's coarse launches deal their few rays out over the waves (one or a few rays per wave) like a single frame's do
mkdir -p gpurun_out/tb
timeout 1500 python -m pytest tests -x -q -m gpu -k "parity or batch or slab or multidevice or temporal or superset or edge or config" 2>&1 | grep -E "passed|failed|rror" | tail -2
B="--no-cpu-baseline --no-extra-legs --sequence none --steps 20 --warmup 5 --sustained-steps 200"
for round in 1 2; do
for v in head new; do
  if [ $v = head ]; then L=$PWD/profiles/variants/libbhray_head.so; else L=$PWD/bhusie_amd/libbhray.so; fi
  BHRAY_LIB=$L timeout 300 python bench.py $B > gpurun_out/tb/n1_$v.json 2>/dev/null
  BHRAY_LIB=$L timeout 300 python bench.py $B --frames-in-flight 1 --sustained-steps 0 > gpurun_out/tb/f1_$v.json 2>/dev/null
  for r in 0 3 5; do BHRAY_LIB=$L timeout 300 python bench.py $B --emulate-world 8 --emulate-rank $r > gpurun_out/tb/r${r}_$v.json 2>/dev/null; done
  BHRAY_LIB=$L timeout 300 python bench.py $B --width 3840 --height 2160 --emulate-world 8 --emulate-rank 4 > gpurun_out/tb/k4_$v.json 2>/dev/null
  BHRAY_LIB=$L timeout 300 python bench.py $B --emulate-world 4 --emulate-rank 1 > gpurun_out/tb/q1_$v.json 2>/dev/null
  python -c "
import json
g=lambda n: json.loads(open('gpurun_out/tb/%s_$v.json' % n).read().strip().splitlines()[-1])
print('$v: N=1', g('n1')['value'], g('n1')['sustained']['mrays_per_s'], 'one frame', g('f1')['ms_per_step'], '| ranks 0/3/5 of 8:', [(g('r%d'%r)['ms_per_step'], g('r%d'%r)['sustained']['ms_per_step']) for r in (0,3,5)], '| 4K rank 4/8', g('k4')['ms_per_step'], g('k4')['sustained']['ms_per_step'], '| rank 1/4', g('q1')['ms_per_step'])"
done
done
