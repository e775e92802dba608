#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for fif in 16 20 24 28; do for S in 0 2; do
  echo -n "fif $fif S $S: "
  a=$(timeout 300 python bench.py --no-cpu-baseline --min-seconds 0.6 --frames-in-flight $fif --speculative-levels $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.6 --frames-in-flight $fif --speculative-levels $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "long $a short $b"
done; done
} > gpurun_out/exp42.log 2>&1
