cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s41; O=gpurun_out/s41
for i in 1 2 3 4 5 6; do
timeout 900 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_repartition.py tests/test_gpu_multirank.py tests/test_gpu_slabs.py tests/test_gpu_handoff.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2
done | tee $O/repeat.txt
