#!/bin/bash
mkdir -p gpurun_out/ms
for v in head sA sB sC sD; do
  L=$PWD/profiles/variants/libbhray_$v.so
  BHRAY_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --sequence none --workload mesh --sustained-steps 200 > gpurun_out/ms/mesh_$v.json 2>/dev/null
  python -c "
import json; a=json.loads(open('gpurun_out/ms/mesh_$v.json').read().strip().splitlines()[-1]); print('$v', a['value'], a['sustained']['mrays_per_s'])"
done
