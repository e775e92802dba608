cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s50; O=gpurun_out/s50
V=$GRAFT_REPO_ROOT/profiles/variants
for r in 1 2 3; do for lib in $V/libbhray_nostride.so ""; do for rk in 0 3; do
  BHRAY_LIB=$lib timeout 300 python bench.py --emulate-world 8 --emulate-rank $rk --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib##*/}', 'rank $rk of 8', d['steps'], d['value'], d['ms_per_step'], d['config']['frames_per_batch'])"
done; done; done 2>&1 | tee $O/emu.txt
