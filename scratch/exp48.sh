#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for fpb in 1 2 4; do
  echo -n "fpb $fpb: "
  a=$(timeout 300 python bench.py --no-cpu-baseline --min-seconds 0.8 --frames-per-batch $fpb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.8 --frames-per-batch $fpb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "long $a short $b"
done
} > gpurun_out/exp48.log 2>&1
