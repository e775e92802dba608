cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s43
export GPU_MAX_HW_QUEUES=64
timeout 1500 python profiles/jobs/r5_soak.py 20000 2>&1 | grep -vE "NCCL|RCCL" | tail -5 | tee gpurun_out/s43/soak.txt
