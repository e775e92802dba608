#!/bin/bash
# 3840x2160 over 8 partitions (balanced slabs + feedback), 20-frame blocks, every rank emulated on one GPU: the N = 1 / slowest-rank ratio;
# one feedback round against two (usage: r4_4k_repeat.sh), two repetitions each
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/k4; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0 --sequence none --width 3840 --height 2160 --steps 20 --warmup 5"
for rep in 1 2; do
  timeout 600 python bench.py $B > $OUT/n1_$rep.json 2>/dev/null
  for rounds in 1 2 3; do
  for r in 0 1 2 3 4 5 6 7; do timeout 600 python bench.py $B --partition-feedback-rounds $rounds --emulate-world 8 --emulate-rank $r > $OUT/r${r}_${rounds}_$rep.json 2>/dev/null; done
  python - <<P
import json
g=lambda n: json.loads(open("gpurun_out/k4/%s.json" % n).read().strip().splitlines()[-1])["ms_per_step"]
n1=g("n1_$rep"); rs=[g("r%d_${rounds}_$rep"%r) for r in range(8)]
print("rep $rep rounds $rounds: N=1", n1, "ranks", rs, "slowest", max(rs), "mean", round(sum(rs)/8,5), "scaling", round(n1/max(rs),3))
P
  done
done
