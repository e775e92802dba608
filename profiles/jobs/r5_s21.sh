cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s21
export GPU_MAX_HW_QUEUES=64
V=$GRAFT_REPO_ROOT/profiles/variants
for r in 1 2; do
for lib in "" $V/libbhray_m_cold6.so $V/libbhray_m_cold6ww.so $V/libbhray_m_cold6wwlds.so $V/libbhray_m_cold6st4.so; do
  BHRAY_LIB=$lib timeout 200 python profiles/jobs/r5_mesh_variants.py 2>&1 | grep "mesh "
done; done | tee gpurun_out/s21/mesh_cold.txt
