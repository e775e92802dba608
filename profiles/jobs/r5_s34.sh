cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s34; O=gpurun_out/s34
V=$GRAFT_REPO_ROOT/profiles/variants
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -5
for r in 1 2; do for lib in $V/libbhray_prev.so ""; do for wl in "" "--workload mesh"; do for st in "--steps 400 --warmup 32"; do
  BHRAY_LIB=$lib timeout 300 python bench.py $st $wl --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', '$wl', d['steps'], d['value'], d['ms_per_step'])"
done; done; done; done 2>&1 | tee $O/regress.txt
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -5
