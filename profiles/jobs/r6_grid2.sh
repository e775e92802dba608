cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/grid2; mkdir -p $OUT; rm -f $OUT/ab.txt
run() { env BHRAY_TRACE_GRID=$1 timeout 300 python bench.py $3 --no-cpu-baseline --no-extra-legs --sustained-steps 0 --steps $2 --warmup 5 --min-seconds 1.5 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid=$1 $3 steps=$2', d['value'], d['ms_per_step'])" >> $OUT/ab.txt; }
for rnd in 1 2; do for g in 0 192 256 320 384 448; do run $g 20 ""; run $g 400 ""; done; done
cat $OUT/ab.txt
