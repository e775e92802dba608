#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for fif in 20 24 28; do
  echo -n "fif $fif: "
  a=$(timeout 300 python bench.py --no-cpu-baseline --min-seconds 0.6 --frames-in-flight $fif 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.6 --frames-in-flight $fif 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "long $a short $b"
done
echo -n "HWQ unset-like (4), fif 20: "
GPU_MAX_HW_QUEUES=4 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
echo -n "HWQ 8, fif 20: "
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multidevice.py tests/test_gpu_temporal.py -x -q 2>&1 | grep -E "passed|failed"
} > gpurun_out/exp43.log 2>&1
