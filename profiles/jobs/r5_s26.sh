cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s26; O=gpurun_out/s26
python profiles/jobs/r5_mesh_latency.py 2>&1 | grep -v "^$" | tee $O/mesh_latency.txt
BHRAY_TRACE_DENSE=1 python profiles/jobs/r5_mesh_latency.py 2>&1 | grep -E "wall" | tee -a $O/mesh_latency.txt
