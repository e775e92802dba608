#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for lib in bhusie_amd/libbhray.so scratch/variants/libbhray_unroll2.so scratch/variants/libbhray_unroll4.so; do
  echo "LIB $lib"
  BHRAY_LIB=$lib python scratch/exp24.py
  BHRAY_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['latency_ms_one_frame_in_flight_by_mode'], d['roofline']['isolated']['level_trace_ms'])"
done
} > gpurun_out/exp41.log 2>&1
