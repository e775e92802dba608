#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh_stack.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py tests/test_gpu_multidevice.py -q -k "mesh or bvh or chain or ring or config2 or fuzz or variant" > gpurun_out/s6/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/s6/pytest.log | tail -2
echo "== mesh A/B"
bash profiles/jobs/r4_ab_mesh.sh 2 default inl5 pool2_w6 pool2_s32 pool2_s56 pool2_d24
