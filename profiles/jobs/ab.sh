# A/B of kernel build variants in one session: default, then each profiles/variants/libbhray_<v>.so, three rounds, the driver's
# command (20-frame blocks, 2 s of them) and 400-frame blocks.  usage: bash profiles/jobs/ab.sh v1 v2 ...
cd $GRAFT_REPO_ROOT
one() {  # $1 = label, $2 = lib or ""
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    BHRAY_LIB=$2 timeout 300 python bench.py $cfg --no-extra-legs --no-cpu-baseline --min-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2 3; do
  one default ""
  for v in "$@"; do one $v $GRAFT_REPO_ROOT/profiles/variants/libbhray_$v.so; done
done
