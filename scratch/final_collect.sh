#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash profiles/collect.sh r02 > gpurun_out/collect.log 2>&1
bash profiles/bench_lines.sh r02 > gpurun_out/bench_lines.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/final_pytest.log
