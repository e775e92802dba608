cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s48
export GPU_MAX_HW_QUEUES=64
timeout 1500 python profiles/jobs/r5_soak.py 8000 worst-first mesh 2>&1 | grep -E "^{|Error|error|assert" | tee gpurun_out/s48/soak_mesh.txt
