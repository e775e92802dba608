cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s39; O=gpurun_out/s39
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multidevice.py tests/test_gpu_repartition.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/tests.txt
