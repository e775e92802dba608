#!/bin/bash
# The drop-in shim with two frames in flight: the latency build's persistent grid fills every wave slot (4 blocks per CU), so frame i + 1's first launch waits for frame i's last - do smaller grids / the dense build overlap better?
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/r6_dropin; mkdir -p $OUT
run() { env $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'block', d['ms_per_step'], 'lat', d['latency_ms_one_frame_in_flight_by_mode']['ladder_speculative_levels_2'], d['latency_ms_one_frame_in_flight_by_mode']['ladder_speculative_levels_3'], 'dropin', {k: v['ms_per_frame'] for k, v in d['dropin']['legs'].items()})" >> $OUT/ab.txt; }
for rnd in 1 2; do
run base X=1; run bpc2 BHRAY_TRACE_BLOCKS_PER_CU=2; run bpc3 BHRAY_TRACE_BLOCKS_PER_CU=3; run dense BHRAY_TRACE_DENSE=1
done
