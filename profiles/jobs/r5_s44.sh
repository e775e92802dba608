cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s44
export GPU_MAX_HW_QUEUES=64
timeout 1500 python profiles/jobs/r5_soak.py 8000 worst-first 2>&1 | grep "^{" | tee gpurun_out/s44/soak_worst_first.txt
