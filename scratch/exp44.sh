#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for cfg in "12 24" "12 23" "12 22" "24 24"; do
  set -- $cfg
  echo -n "HWQ $1 fif $2: "
  b=$(GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.4 --frames-in-flight $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_blocks']['block_ms'])")
  echo "short $b"
done
echo "steps 24, fif 24:"; timeout 300 python bench.py --no-cpu-baseline --steps 24 --warmup 5 --min-seconds 0.4 --frames-in-flight 24 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_blocks']['block_ms'])"
echo "steps 20, warmup 4, fif 24:"; timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 4 --min-seconds 0.4 --frames-in-flight 24 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_blocks']['block_ms'])"
} > gpurun_out/exp44.log 2>&1
