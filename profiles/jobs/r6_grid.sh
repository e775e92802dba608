#!/bin/bash
# BHRAY_TRACE_GRID (absolute number of persistent trace blocks; 256 CUs) on the final kernels
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/grid; mkdir -p $OUT; rm -f $OUT/ab.txt
run() { # grid steps extra
  env BHRAY_TRACE_GRID=$1 timeout 300 python bench.py $3 --no-cpu-baseline --no-extra-legs --sustained-steps 0 --steps $2 --warmup 5 --min-seconds 1.5 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid=$1 $3 steps=$2', d['value'], d['ms_per_step'])" >> $OUT/ab.txt
}
for rnd in 1 2; do
for g in 0 320 384 448 512 576 640; do run $g 20 ""; run $g 400 ""; done
for g in 0 256 320 384 448; do run $g 20 "--integrator euler"; run $g 400 "--integrator euler"; done
done
cat $OUT/ab.txt
