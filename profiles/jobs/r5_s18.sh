cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s18
export GPU_MAX_HW_QUEUES=64
V=$GRAFT_REPO_ROOT/profiles/variants
for r in 1 2; do
for lib in "" $V/libbhray_m_ni5.so $V/libbhray_m_ni6.so $V/libbhray_m_ni6cold.so $V/libbhray_m_ni8.so; do
  BHRAY_LIB=$lib timeout 200 python profiles/jobs/r5_mesh_variants.py 2>&1 | grep "mesh "
done; done | tee gpurun_out/s18/mesh_noinline.txt
