#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for rep in 1 2 3 4; do
for lib in scratch/variants/libbhray_prev.so bhusie_amd/libbhray.so; do
  echo "LIB $lib"
  BHRAY_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --min-seconds 1.0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps'], d['value'], d['timed_blocks']['block_ms'])"
  BHRAY_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 1.0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps'], d['value'], d['timed_blocks']['block_ms'])"
done; done
} > gpurun_out/exp32.log 2>&1
