#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for rep in 1 2; do for n in 0 2 4 6; do
  export BHRAY_COARSE_STREAMS=$n
  echo -n "coarse streams $n: "
  a=$(timeout 300 python bench.py --no-cpu-baseline --min-seconds 0.8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_blocks']['block_ms']['min'])")
  echo "long $a short $b"
done; done
BHRAY_COARSE_STREAMS=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multidevice.py tests/test_gpu_edge_cases.py -x -q 2>&1 | grep -E "passed|failed"
} > gpurun_out/exp62.log 2>&1
