#!/bin/bash
# mesh variant: the cull phase split from the traversal phase (BHRAY_FLAT_SPLIT) against the build before it
mkdir -p gpurun_out/ms
timeout 1200 python -m pytest tests -x -q -m gpu -k "mesh or config2 or bvh or sah or model" > gpurun_out/ms/test.log 2>&1; echo "mesh tests rc=$?"; grep -E "passed|failed" gpurun_out/ms/test.log | tail -2
for round in 1 2; do
for v in head new; do
  if [ $v = head ]; then L=$PWD/profiles/variants/libbhray_head.so; else L=$PWD/bhusie_amd/libbhray.so; fi
  BHRAY_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --sequence none --workload mesh --sustained-steps 200 > gpurun_out/ms/mesh_$v.json 2>/dev/null
  BHRAY_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --sequence none --workload mesh --frames-in-flight 1 --sustained-steps 0 > gpurun_out/ms/mesh_f1_$v.json 2>/dev/null
  python -c "
import json; a=json.loads(open('gpurun_out/ms/mesh_$v.json').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/ms/mesh_f1_$v.json').read().strip().splitlines()[-1]); print('$v', a['value'], a['sustained']['mrays_per_s'], 'one frame', b['ms_per_step'])"
done
done
