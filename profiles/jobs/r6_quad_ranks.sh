#!/bin/bash
# The quad march for a RANK of an 8-way partition (emulated on one GPU, before the gather): the driver's 20-frame blocks, a long block and one
# frame at a time, BHRAY_QUAD=0 against the default, ranks 0 (sky slab), 3 and 4 (the hole's slabs).  A/B alternating within one run.
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/r6_quad_ranks; mkdir -p $OUT
run() { # label env args
  env $2 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --sustained-steps 0 --emulate-world 8 $3 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'fpb', d['config'].get('frames_per_batch'))" >> $OUT/ab.txt
}
for rnd in 1 2; do
for r in 0 3 4; do
  run "rank$r short quad0" BHRAY_QUAD=0 "--emulate-rank $r --steps 20 --warmup 5 --min-seconds 1"
  run "rank$r short quadD" X=1 "--emulate-rank $r --steps 20 --warmup 5 --min-seconds 1"
  run "rank$r long  quad0" BHRAY_QUAD=0 "--emulate-rank $r --steps 1000 --warmup 40 --min-seconds 0.3"
  run "rank$r long  quadD" X=1 "--emulate-rank $r --steps 1000 --warmup 40 --min-seconds 0.3"
  run "rank$r one   quad0" BHRAY_QUAD=0 "--emulate-rank $r --steps 20 --warmup 5 --min-seconds 0.3 --frames-in-flight 1 --frames-per-batch 1"
  run "rank$r one   quadD" X=1 "--emulate-rank $r --steps 20 --warmup 5 --min-seconds 0.3 --frames-in-flight 1 --frames-per-batch 1"
done; done
