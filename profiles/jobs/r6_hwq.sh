#!/bin/bash
# GPU_MAX_HW_QUEUES and frame slots on the final kernels: does the number of hardware queues the 22 streams map onto change how full the device gets?
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/hwq; mkdir -p $OUT; rm -f $OUT/ab.txt
run() { # queues slots steps
  env GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --frames-in-flight $2 --no-cpu-baseline --no-extra-legs --sustained-steps 0 --steps $3 --warmup 5 --min-seconds 1.5 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues=$1 slots=$2 steps=$3', d['value'], d['ms_per_step'])" >> $OUT/ab.txt
}
for q in 64 32 24 16 8 4; do run $q 22 20; run $q 22 400; done
for s in 12 16 32 44; do run 64 $s 20; run 64 $s 400; done
cat $OUT/ab.txt
