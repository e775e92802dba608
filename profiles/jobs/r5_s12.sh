cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s12; O=gpurun_out/s12
export GPU_MAX_HW_QUEUES=64
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_literal.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -3
one() {  # label lib workload-args
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    BHRAY_LIB=$2 timeout 300 python bench.py $cfg $3 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2 3; do
  for wl in "" "--integrator euler" "--workload mesh"; do
    one prev $GRAFT_REPO_ROOT/profiles/variants/libbhray_prev.so "$wl"
    one folded "" "$wl"
  done
done 2>&1 | tee $O/ab_culls.txt
