#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python scratch/exp24.py > gpurun_out/exp39.log 2>&1
FINE=1 BHRAY_LIB=scratch/variants/libbhray_prof2.so python scratch/exp39.py >> gpurun_out/exp39.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | grep -E "passed|failed" >> gpurun_out/exp39.log
