cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g6
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed|Error|error" | head -5
for bpc in 4 2 1; do
BHRAY_TRACE_BLOCKS_PER_CU=$bpc BHRAY_LIB=$GRAFT_REPO_ROOT/profiles/variants/libbhray_prof.so timeout 300 python profiles/fused_profile.py 2 > gpurun_out/g6/prof_bpc$bpc.json 2> gpurun_out/g6/err_$bpc.txt
cat gpurun_out/g6/prof_bpc$bpc.json
done
