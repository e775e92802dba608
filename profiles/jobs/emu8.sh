# N=8 emulated rank 3, 20-frame blocks: which knobs shorten the block? (dense/latency build, speculative levels, frames per batch, slots)
cd $GRAFT_REPO_ROOT
run() { # label, env, args
  env $2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --emulate-world 8 --emulate-rank 3 --min-seconds 1 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config'].get('frames_per_batch'), d['config'].get('frames_in_flight'), d['config'].get('speculative_levels'))"
}
run base X=1 ""
run latency_build BHRAY_TRACE_DENSE=0 ""
run dense_build BHRAY_TRACE_DENSE=1 ""
run spec3 X=1 "--speculative-levels 3"
run spec3_latency BHRAY_TRACE_DENSE=0 "--speculative-levels 3"
for b in 4 5 7 10 20; do run fpb$b X=1 "--frames-per-batch $b"; run fpb${b}_lat BHRAY_TRACE_DENSE=0 "--frames-per-batch $b"; done
for s in 2 4 8; do run slots$s X=1 "--frames-in-flight $s"; done
run base X=1 ""
