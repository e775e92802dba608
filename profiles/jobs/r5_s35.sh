cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s35; O=gpurun_out/s35
V=$GRAFT_REPO_ROOT/profiles/variants
python profiles/jobs/r5_mesh_sizes.py 2>&1 | grep subdivision | tee $O/mesh_sizes.txt
BHRAY_LIB=$V/libbhray_l_longest.so python profiles/jobs/r5_mesh_sizes.py 2>&1 | grep subdivision | tee -a $O/mesh_sizes.txt
