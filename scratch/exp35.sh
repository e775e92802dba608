#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for w in 8; do
for fpb in 10 4; do
for dense in 0 1; do
  export BHRAY_TRACE_DENSE=$dense
  echo -n "world $w fpb $fpb dense $dense: "
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.3 --emulate-world $w --emulate-rank 0 --frames-per-batch $fpb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['frames_per_batch'], d['timed_blocks']['block_ms']['median'])"
done; done; done
unset BHRAY_TRACE_DENSE
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/exp35_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/exp35_pytest.log
} > gpurun_out/exp35.log 2>&1
