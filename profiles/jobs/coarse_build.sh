# the trace build of the launches below the last level (BHRAY_COARSE_BUILD: 0 latency build, 1 dense) against the ctx-wide choice
cd $GRAFT_REPO_ROOT
run() { # label, env, args
  env $2 timeout 200 python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 2 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['steps'], d['ms_per_step'], d['value'])"
}
for r in 1 2; do
run n1_default X=1 "--steps 20 --warmup 5"
run n1_coarse_latency BHRAY_COARSE_BUILD=0 "--steps 20 --warmup 5"
run n1_default X=1 "--steps 400 --warmup 32"
run n1_coarse_latency BHRAY_COARSE_BUILD=0 "--steps 400 --warmup 32"
run n8_default X=1 "--steps 20 --warmup 5 --emulate-world 8 --emulate-rank 3"
run n8_coarse_latency BHRAY_COARSE_BUILD=0 "--steps 20 --warmup 5 --emulate-world 8 --emulate-rank 3"
run n8_all_latency BHRAY_TRACE_DENSE=0 "--steps 20 --warmup 5 --emulate-world 8 --emulate-rank 3"
run n4_default X=1 "--steps 20 --warmup 5 --emulate-world 4 --emulate-rank 1"
run n4_coarse_latency BHRAY_COARSE_BUILD=0 "--steps 20 --warmup 5 --emulate-world 4 --emulate-rank 1"
run n4_all_latency BHRAY_TRACE_DENSE=0 "--steps 20 --warmup 5 --emulate-world 4 --emulate-rank 1"
done
