cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s22
export GPU_MAX_HW_QUEUES=64
V=$GRAFT_REPO_ROOT/profiles/variants
for L in $V/libbhray_m_park6.so $V/libbhray_m_park5.so; do
BHRAY_LIB=$L timeout 900 python -m pytest tests/test_gpu_bvh_stack.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_configs.py -x -q -m gpu -k "mesh or bvh or config2 or stack or chain or depth" 2>&1 | grep -E "passed|failed|Error" | tail -3
done
for r in 1 2; do
for lib in "" $V/libbhray_m_park5.so $V/libbhray_m_park6.so $V/libbhray_m_park6ww.so $V/libbhray_m_park8.so; do
  BHRAY_LIB=$lib timeout 200 python profiles/jobs/r5_mesh_variants.py 2>&1 | grep "mesh "
done; done | tee gpurun_out/s22/mesh_park.txt
