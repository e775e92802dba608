#!/bin/bash
# wave priority for a RANK of an 8-way partition (emulated, before the gather): BHRAY_PRIO=0 / 1 forced, the driver's 20-frame blocks, 1000-frame blocks, one frame at a time
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/r6_prio_ranks; mkdir -p $OUT
run() { env $2 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --sustained-steps 0 --emulate-world 8 $3 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'fpb', d['config'].get('frames_per_batch'))" >> $OUT/ab.txt; }
for rnd in 1 2; do for r in 0 3; do
  run "rank$r short prio0" BHRAY_PRIO=0 "--emulate-rank $r --steps 20 --warmup 5 --min-seconds 1"
  run "rank$r short prio1" BHRAY_PRIO=1 "--emulate-rank $r --steps 20 --warmup 5 --min-seconds 1"
  run "rank$r long  prio0" BHRAY_PRIO=0 "--emulate-rank $r --steps 1000 --warmup 40 --min-seconds 0.3"
  run "rank$r long  prio1" BHRAY_PRIO=1 "--emulate-rank $r --steps 1000 --warmup 40 --min-seconds 0.3"
  run "rank$r one   prio0" BHRAY_PRIO=0 "--emulate-rank $r --steps 20 --warmup 5 --min-seconds 0.3 --frames-in-flight 1 --frames-per-batch 1"
  run "rank$r one   prio1" BHRAY_PRIO=1 "--emulate-rank $r --steps 20 --warmup 5 --min-seconds 0.3 --frames-in-flight 1 --frames-per-batch 1"
done; done
