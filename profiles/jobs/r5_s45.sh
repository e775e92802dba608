cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s45; O=gpurun_out/s45
export GPU_MAX_HW_QUEUES=64
for r in 1 2; do for g in "" 320 384 448; do
  for st in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
  BHRAY_TRACE_GRID=$g timeout 300 python bench.py $st --integrator euler --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid=$g', 'euler N=1', d['steps'], d['value'], d['ms_per_step'])"
  BHRAY_TRACE_GRID=$g timeout 300 python bench.py $st --integrator euler --workload mesh --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid=$g', 'euler mesh N=1', d['steps'], d['value'], d['ms_per_step'])"
  BHRAY_TRACE_GRID=$g timeout 300 python bench.py $st --integrator euler --emulate-world 8 --emulate-rank 3 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid=$g', 'euler rank 3 of 8', d['steps'], d['value'], d['ms_per_step'])"
  done
done; done 2>&1 | tee $O/euler_grid.txt
