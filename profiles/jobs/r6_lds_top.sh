#!/bin/bash
# R6.8: the top of the BVH staged in LDS per wave (-DBHRAY_BVH_LDS_TOP=128) against the shipped mesh kernels: parity, blocks, one frame at a time
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/r6_top; mkdir -p $OUT
V=$PWD/profiles/variants/libbhray_top128.so
BHRAY_LIB=$V python -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh_stack.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py -q -m gpu -k "mesh or model or bvh or config2 or triangle" > $OUT/pytest.txt 2>&1
run() { env BHRAY_LIB=$2 timeout 300 python bench.py --workload mesh --no-cpu-baseline --no-extra-legs --sustained-steps 0 --warmup 5 --min-seconds 1.5 $3 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 |$3|', d['value'], d['ms_per_step'])" >> $OUT/blocks.txt; }
for rnd in 1 2 3; do for a in "--steps 20" "--steps 400" "--steps 20 --integrator euler" "--steps 400 --integrator euler"; do
  run base $PWD/bhusie_amd/libbhray.so "$a"; run top128 $V "$a"
done; done
python profiles/jobs/r6_lat_ab.py --mesh --no-timing bhusie_amd/libbhray.so profiles/variants/libbhray_top128.so 3 > $OUT/lat.txt 2>&1
