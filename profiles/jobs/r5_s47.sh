cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s47; O=gpurun_out/s47
for r in 1 2; do for g in "" 320 384 448 640; do for st in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
  BHRAY_TRACE_GRID=$g timeout 300 python bench.py $st --workload mesh --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid=$g', 'rk mesh N=1', d['steps'], d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/mesh_grid.txt
