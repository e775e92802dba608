cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s49; O=gpurun_out/s49
V=$GRAFT_REPO_ROOT/profiles/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_bvh_stack.py tests/test_gpu_temporal.py tests/test_gpu_superset.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 | tee $O/tests.txt
for r in 1 2; do for lib in $V/libbhray_nostride.so ""; do
  echo "== ${lib##*/}"; BHRAY_LIB=$lib python profiles/jobs/r5_lat.py 2>&1 | grep wall; BHRAY_LIB=$lib python profiles/jobs/r5_lat8.py 2>&1 | grep wall
done; done 2>&1 | tee $O/latency.txt
for r in 1 2; do for lib in $V/libbhray_nostride.so ""; do
  BHRAY_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', 'default bench', d['value'], d['ms_per_step'], 'latency', d['latency_ms_one_frame_in_flight'], d['latency_ms_one_frame_in_flight_by_mode'], 'dropin', json.dumps(d.get('dropin'))[:300])"
done; done 2>&1 | tee $O/bench.txt
