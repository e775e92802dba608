cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s15; O=gpurun_out/s15
export GPU_MAX_HW_QUEUES=64
one() {  # label lib workload-args
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    BHRAY_LIB=$2 timeout 300 python bench.py $cfg $3 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2 3; do
  for wl in "" "--integrator euler"; do
    one default "" "$wl"
    one noguard $GRAFT_REPO_ROOT/profiles/variants/libbhray_noguard.so "$wl"
  done
done 2>&1 | tee $O/ab_noguard.txt
