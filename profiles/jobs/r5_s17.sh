cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s17
export GPU_MAX_HW_QUEUES=64
V=$GRAFT_REPO_ROOT/profiles/variants
for r in 1 2; do
for lib in "" $V/libbhray_m_notrav.so $V/libbhray_m_w6.so $V/libbhray_m_w4.so $V/libbhray_m_coldlds.so $V/libbhray_m_flat1.so $V/libbhray_m_notrav6.so $V/libbhray_m_notrav6cold.so; do
  BHRAY_LIB=$lib timeout 200 python profiles/jobs/r5_mesh_variants.py 2>&1 | grep "mesh "
done; done | tee gpurun_out/s17/mesh_variants.txt
