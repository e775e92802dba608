#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for fpb in 10 5 20; do
  echo "fpb $fpb"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.3 --emulate-world 8 --emulate-rank 0 --frames-per-batch $fpb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'issue', d['host_issue_ms_per_step'], d['pass_ms'], d['timed_blocks'])"
done
} > gpurun_out/exp36.log 2>&1
