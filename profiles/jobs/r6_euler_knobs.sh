#!/bin/bash
# the dense Euler kernel with its own batch length (steps between the phases) and refill threshold - builds under profiles/variants/
# (make OUT=... EXTRA=-DBHRAY_REL_BATCH_EULER_DENSE=n / -DBHRAY_REFILL_MIN_EULER_DENSE=n), the driver's 20-frame blocks and 400-frame blocks, alternating, 2 rounds.
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/r6_euler; mkdir -p $OUT
run() { # label lib steps
  env BHRAY_LIB=$2 timeout 300 python bench.py --integrator euler --no-cpu-baseline --no-extra-legs --sustained-steps 0 --steps $3 --warmup 5 --min-seconds 1.5 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 steps=$3', d['value'], d['ms_per_step'])" >> $OUT/ab.txt
}
for rnd in 1 2; do
  for v in base b8 b24 b32 b48 r8 r24 r32 b32r24; do
    lib=$PWD/profiles/variants/libbhray_e_$v.so; [ $v = base ] && lib=$PWD/bhusie_amd/libbhray.so
    run $v $lib 20; run $v $lib 400
  done
done
