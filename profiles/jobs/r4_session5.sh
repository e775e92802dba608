#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/s5/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/s5/pytest.log | tail -2; grep -E "^FAILED" gpurun_out/s5/pytest.log | head
echo "== mesh A/B"
bash profiles/jobs/r4_ab_mesh.sh 2 default inl5 pool_w4 pool_s16 pool_s48 pool_d16 pool_d4 pool_a12
