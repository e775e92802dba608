#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for rep in 1 2; do for fif in 20 21 22 24; do
  echo -n "fif $fif: "
  a=$(timeout 300 python bench.py --no-cpu-baseline --min-seconds 0.8 --frames-in-flight $fif 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.8 --frames-in-flight $fif 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_blocks']['block_ms']['max'])")
  echo "long $a short $b"
done; done
} > gpurun_out/exp45.log 2>&1
