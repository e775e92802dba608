#!/bin/bash
mkdir -p gpurun_out/ro
B="--no-cpu-baseline --no-extra-legs --sequence none --steps 20 --warmup 5 --sustained-steps 200"
for round in 1 2; do
for v in norows new; do
  if [ $v = norows ]; then L=$PWD/profiles/variants/libbhray_norows.so; else L=$PWD/bhusie_amd/libbhray.so; fi
  BHRAY_LIB=$L timeout 300 python bench.py $B --workload mesh > gpurun_out/ro/m_$v.json 2>/dev/null
  BHRAY_LIB=$L timeout 300 python bench.py $B --workload mesh --frames-in-flight 1 --sustained-steps 0 > gpurun_out/ro/mf_$v.json 2>/dev/null
  BHRAY_LIB=$L timeout 300 python bench.py $B --integrator euler > gpurun_out/ro/e_$v.json 2>/dev/null
  BHRAY_LIB=$L timeout 300 python bench.py $B --width 3840 --height 2160 > gpurun_out/ro/k_$v.json 2>/dev/null
  BHRAY_LIB=$L timeout 300 python bench.py $B > gpurun_out/ro/n_$v.json 2>/dev/null
  python -c "
import json
g=lambda n: json.loads(open('gpurun_out/ro/%s_$v.json' % n).read().strip().splitlines()[-1])
print('$v: mesh', g('m')['value'], g('m')['sustained']['mrays_per_s'], 'one frame', g('mf')['ms_per_step'], '| euler', g('e')['value'], '| 4K', g('k')['value'], '| default', g('n')['value'], g('n')['sustained']['mrays_per_s'])"
done
done
