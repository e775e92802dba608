#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for lib in bhusie_amd/libbhray.so scratch/variants/libbhray_meshlds.so; do
  echo "LIB $lib"
  for i in 1 2; do BHRAY_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --workload mesh --steps 200 --warmup 32 --min-seconds 0.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v for k,v in d['latency_ms_one_frame_in_flight_by_mode'].items() if k!='note'})"; done
done
BHRAY_LIB=scratch/variants/libbhray_meshlds.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mesh" 2>&1 | grep -E "passed|failed"
} > gpurun_out/exp54.log 2>&1
