#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
bash profiles/collect.sh r02 > gpurun_out/collect.log 2>&1
bash profiles/bench_lines.sh r02 > gpurun_out/bench_lines.log 2>&1
python profiles/latency_experiments.py > gpurun_out/latency_experiments.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/final_pytest.log
tail -1 gpurun_out/smoke.log
