cd $GRAFT_REPO_ROOT
one() { for cfg in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do BHRAY_LIB=$2 timeout 300 python bench.py $cfg --workload mesh --no-extra-legs --no-cpu-baseline --min-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['steps'], d['value'], d['ms_per_step'])"; done; }
for r in 1 2; do one default ""; for v in "$@"; do one $v $GRAFT_REPO_ROOT/profiles/variants/libbhray_$v.so; done; done
