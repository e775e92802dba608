cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s40; O=gpurun_out/s40
V=$GRAFT_REPO_ROOT/profiles/variants
for r in 1 2 3; do for lib in "" $V/libbhray_e_w8.so $V/libbhray_e_w7.so; do for wl in "--integrator euler"; do for st in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
  BHRAY_LIB=$lib timeout 300 python bench.py $st $wl --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', '$wl', d['steps'], d['value'], d['ms_per_step'])"
done; done; done; done 2>&1 | tee $O/euler_waves.txt
