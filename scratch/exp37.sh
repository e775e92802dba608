#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
for dense in auto 1; do
  if [ $dense = auto ]; then unset BHRAY_TRACE_DENSE; else export BHRAY_TRACE_DENSE=$dense; fi
  for res in "1920 1080" "3840 2160"; do
  set -- $res
  echo "dense $dense $1x$2"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.2 --width $1 --height $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['latency_ms_one_frame_in_flight_by_mode'], d['roofline']['isolated']['level_trace_ms'])"
  done
done
} > gpurun_out/exp37.log 2>&1
