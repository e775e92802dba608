cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s30; O=gpurun_out/s30
export GPU_MAX_HW_QUEUES=64
V=$GRAFT_REPO_ROOT/profiles/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_bvh_stack.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 | tee $O/tests.txt
for r in 1 2; do for lib in $V/libbhray_prev.so "" $V/libbhray_l_strided64.so; do
  echo "== ${lib##*/}"; BHRAY_LIB=$lib python profiles/jobs/r5_lat.py 2>&1 | grep wall
done; done 2>&1 | tee $O/latency_strided.txt
for lib in $V/libbhray_prev.so "" $V/libbhray_l_strided64.so; do for r in 1 2; do
  BHRAY_LIB=$lib timeout 300 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', '8 partitions one GPU', d['steps'], d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $O/eight.txt
