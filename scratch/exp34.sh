#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for w in 8 4 2; do
for fpb in -1 4 5 10 20; do
for dense in auto 0 1; do
  if [ $dense = auto ]; then unset BHRAY_TRACE_DENSE; else export BHRAY_TRACE_DENSE=$dense; fi
  echo -n "world $w fpb $fpb dense $dense: "
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.3 --emulate-world $w --emulate-rank 0 --frames-per-batch $fpb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['frames_per_batch'], d['timed_blocks']['block_ms']['median'])"
done; done; done
} > gpurun_out/exp34.log 2>&1
