#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
python scratch/exp24.py
timeout 600 python bench.py --no-cpu-baseline --min-seconds 1.0
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 1.0
timeout 600 python bench.py --no-cpu-baseline --workload mesh
} > gpurun_out/exp40.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/exp40_pytest.log 2>&1
