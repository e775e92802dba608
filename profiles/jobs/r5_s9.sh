cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s9; O=gpurun_out/s9
export GPU_MAX_HW_QUEUES=64
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ladder or level0" 2>&1 | tail -3
BHRAY_FRONT_PRIORITY=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_superset.py tests/test_gpu_temporal.py -x -q -m gpu 2>&1 | tail -3
one() {  # label env fif args
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    env $2 timeout 300 python bench.py $cfg --frames-in-flight $3 $4 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'fif $3', '$4', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2; do
  one base BHRAY_FRONT_PRIORITY=0 22 ""
  one base BHRAY_FRONT_PRIORITY=0 11 ""
  one front BHRAY_FRONT_PRIORITY=1 11 ""
  one front BHRAY_FRONT_PRIORITY=1 8 ""
  one base BHRAY_FRONT_PRIORITY=0 11 "--emulate-world 8 --emulate-rank 3 --partition stripes"
  one front BHRAY_FRONT_PRIORITY=1 11 "--emulate-world 8 --emulate-rank 3 --partition stripes"
  one base BHRAY_FRONT_PRIORITY=0 22 "--emulate-world 8 --emulate-rank 3 --partition stripes"
done 2>&1 | tee $O/ab_front_priority.txt
