cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s52; O=gpurun_out/s52
V=$GRAFT_REPO_ROOT/profiles/variants
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_configs.py tests/test_gpu_literal.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 | tee $O/tests.txt
for r in 1 2 3; do for lib in $V/libbhray_before_plane.so ""; do for wl in "--integrator euler" "--integrator euler --workload mesh"; do for st in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
  BHRAY_LIB=$lib timeout 300 python bench.py $st $wl --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', '$wl', d['steps'], d['value'], d['ms_per_step'])"
done; done; done; done 2>&1 | tee $O/euler_plane.txt
for lib in $V/libbhray_before_plane.so ""; do echo "== ${lib##*/}"; BHRAY_LIB=$lib python profiles/jobs/r5_lat.py 2>&1 | grep wall | grep euler; done | tee -a $O/euler_plane.txt
