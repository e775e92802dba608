cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g3
AMD_LOG_LEVEL=1 timeout 300 python profiles/jobs/g3b.py > gpurun_out/g3/out.txt 2> gpurun_out/g3/err.txt
echo "rc=$?" >> gpurun_out/g3/out.txt
cat gpurun_out/g3/out.txt; tail -n 8 gpurun_out/g3/err.txt
if grep -q "rc=0" gpurun_out/g3/out.txt; then
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -q --timeout 300 > gpurun_out/g3/pytest_fused.log 2>&1
tail -n 30 gpurun_out/g3/pytest_fused.log
fi
