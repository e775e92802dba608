cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s33; O=gpurun_out/s33
for r in 1 2 3; do for cb in "" 0; do for wl in "" "--integrator euler" "--workload mesh"; do for st in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
  BHRAY_COARSE_BUILD=$cb timeout 300 python bench.py $st $wl --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coarse_build=$cb', '$wl', d['steps'], d['value'], d['ms_per_step'])"
done; done; done; done 2>&1 | tee $O/coarse.txt
