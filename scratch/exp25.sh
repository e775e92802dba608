#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
python scratch/exp24.py
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_literal.py tests/test_gpu_temporal.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5
} > gpurun_out/exp25.log 2>&1
