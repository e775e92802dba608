cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s46; O=gpurun_out/s46
export GPU_MAX_HW_QUEUES=64
V=$GRAFT_REPO_ROOT/profiles/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 | tee $O/tests.txt
for r in 1 2; do for lib in $V/libbhray_prev.so ""; do for st in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
  BHRAY_LIB=$lib timeout 300 python bench.py $st --integrator euler --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', 'euler N=1', d['steps'], d['value'], d['ms_per_step'])"
  BHRAY_LIB=$lib timeout 300 python bench.py $st --integrator euler --gpus 8 --devices 0,0,0,0,0,0,0,0 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', 'euler 8 partitions one GPU', d['steps'], d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/euler_after.txt
