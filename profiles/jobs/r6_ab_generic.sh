#!/bin/bash
# Generic A/B: the product library against variant builds under profiles/variants/ (libbhray_<v>.so), alternating, ROUNDS rounds,
# the driver's 20-frame blocks and 400-frame blocks.  usage: r6_ab_generic.sh OUTDIR ROUNDS "bench args" v1 v2 ...
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/$1; ROUNDS=$2; ARGS=$3; shift 3
mkdir -p $OUT
run() { # label lib steps
  env BHRAY_LIB=$2 timeout 300 python bench.py $ARGS --no-cpu-baseline --no-extra-legs --sustained-steps 0 --steps $3 --warmup 5 --min-seconds 1.5 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 steps=$3', d['value'], d['ms_per_step'])" >> $OUT/ab.txt
}
for rnd in $(seq 1 $ROUNDS); do
  for v in base "$@"; do
    lib=$PWD/profiles/variants/libbhray_$v.so; [ $v = base ] && lib=$PWD/bhusie_amd/libbhray.so
    run $v $lib 20; run $v $lib 400
  done
done
cat $OUT/ab.txt
