# trace build chosen by the batches in flight at launch time (BHRAY_DYNAMIC_DENSE=thr: dense iff in_flight x batch >= thr x partitions)
cd $GRAFT_REPO_ROOT
run() { env $2 timeout 200 python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 2 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['steps'], d['ms_per_step'], d['value'])"; }
for r in 1 2; do
for cfg in "n1:" "n8:--emulate-world 8 --emulate-rank 3" "n4:--emulate-world 4 --emulate-rank 1" "n2:--emulate-world 2 --emulate-rank 1"; do
  n=${cfg%%:*}; a=${cfg#*:}
  for st in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    run ${n}_static X=1 "$st $a"
    run ${n}_dyn4 BHRAY_DYNAMIC_DENSE=4 "$st $a"
    run ${n}_dyn8 BHRAY_DYNAMIC_DENSE=8 "$st $a"
  done
done
done
