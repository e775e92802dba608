#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for g in 0 384 640 768 1024 1536; do
  if [ $g = 0 ]; then unset BHRAY_TRACE_GRID; else export BHRAY_TRACE_GRID=$g; fi
  echo -n "grid $g: "
  a=$(timeout 300 python bench.py --no-cpu-baseline --min-seconds 0.6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "long $a short $b"
done
} > gpurun_out/exp53.log 2>&1
