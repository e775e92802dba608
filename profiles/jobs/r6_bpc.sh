#!/bin/bash
# persistent trace blocks per CU (BHRAY_TRACE_BLOCKS_PER_CU) on the current build: the driver's 20-frame blocks and 400-frame blocks, RK and Euler
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/bpc; mkdir -p $OUT; rm -f $OUT/ab.txt
run() { # bpc steps extra
  env BHRAY_TRACE_BLOCKS_PER_CU=$1 timeout 300 python bench.py $3 --no-cpu-baseline --no-extra-legs --sustained-steps 0 --steps $2 --warmup 5 --min-seconds 1.5 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bpc=$1 $3 steps=$2', d['value'], d['ms_per_step'])" >> $OUT/ab.txt
}
for rnd in 1 2; do
  for b in 0 2 3 4 6; do run $b 20 ""; run $b 400 ""; done
  for b in 0 2 3 4; do run $b 20 "--integrator euler"; run $b 400 "--integrator euler"; done
done
cat $OUT/ab.txt
