cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s5; O=gpurun_out/s5
export GPU_MAX_HW_QUEUES=64
timeout 900 python -m pytest tests/test_gpu_repartition.py tests/test_gpu_multidevice.py tests/test_gpu_slabs.py tests/test_gpu_temporal.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python profiles/jobs/r5_rebalance_emu.py $O/rebalance_emulated.json 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40
