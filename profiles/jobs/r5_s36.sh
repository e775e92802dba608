cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s36; O=gpurun_out/s36
export PYTHONPATH=$GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/profiles/variants
timeout 1500 python -m pytest tests -x -q -m gpu -k "mesh or bvh or config2 or stack or chain or depth or model or fuzz" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/mesh_tests.txt
for r in 1 2; do for lib in $V/libbhray_prev.so ""; do
  echo "== ${lib##*/}"; BHRAY_LIB=$lib python profiles/jobs/r5_lat.py 2>&1 | grep "wall" | grep mesh
done; done 2>&1 | tee $O/latency_leaf.txt
for r in 1 2 3; do for lib in $V/libbhray_prev.so ""; do for wl in "--workload mesh" "--workload mesh --integrator euler"; do for st in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
  BHRAY_LIB=$lib timeout 300 python bench.py $st $wl --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', '$wl', d['steps'], d['value'], d['ms_per_step'])"
done; done; done; done 2>&1 | tee $O/ab_leaf.txt
