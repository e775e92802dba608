#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for f in 0 0.15 0.25 0.3 0.35 0.4 0.45 0.5; do
  echo -n "qstart $f: "
  BHRAY_QUEUE_START=$f timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v for k,v in d['latency_ms_one_frame_in_flight_by_mode'].items() if k!='note'}, d['roofline']['isolated']['level_trace_ms'])"
done
} > gpurun_out/exp49.log 2>&1
