#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for rep in 1 2; do
for lib in scratch/variants/libbhray_prev.so bhusie_amd/libbhray.so; do
  echo "LIB $lib"
  BHRAY_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --min-seconds 1.0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps'], d['value'], d['timed_blocks']['block_ms'], d['latency_ms_one_frame_in_flight_by_mode'])"
  BHRAY_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 1.0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps'], d['value'], d['timed_blocks']['block_ms'])"
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_temporal.py tests/test_gpu_superset.py -x -q 2>&1 | tail -3
} > gpurun_out/exp33.log 2>&1
