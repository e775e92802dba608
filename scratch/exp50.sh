#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v for k,v in d['latency_ms_one_frame_in_flight_by_mode'].items() if k!='note'})"
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v for k,v in d['latency_ms_one_frame_in_flight_by_mode'].items() if k!='note'})"
timeout 900 python -m pytest tests/test_gpu_temporal.py -x -q 2>&1 | grep -E "passed|failed"
} > gpurun_out/exp50.log 2>&1
