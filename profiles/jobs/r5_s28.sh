cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s28; O=gpurun_out/s28
V=$GRAFT_REPO_ROOT/profiles/variants
BHRAY_LIB=$V/libbhray_l_spread.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_bvh_stack.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 | tee $O/tests.txt
for lib in "" $V/libbhray_l_spread.so $V/libbhray_l_w4.so; do
  echo "== ${lib##*/}"; BHRAY_LIB=$lib python profiles/jobs/r5_lat.py 2>&1 | grep wall
done 2>&1 | tee $O/latency_spread.txt
