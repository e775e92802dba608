# A/B with the latency legs: usage: bash profiles/jobs/ab_lat.sh v1 v2 ...   (default first, three rounds)
cd $GRAFT_REPO_ROOT
one() {
  BHRAY_LIB=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['latency_ms_one_frame_in_flight_by_mode']
print('$1', d['value'], 'S2', m['ladder_speculative_levels_2'], 'S3', m['ladder_speculative_levels_3'], 'temporal', m['temporal_static_camera'], m['temporal_moving_camera'], 'literal', d['literal']['mrays_per_s'])"
  BHRAY_LIB=$2 timeout 300 python bench.py --steps 400 --warmup 32 --no-cpu-baseline --no-extra-legs --min-seconds 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 400-blocks', d['value'])"
  BHRAY_LIB=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --min-seconds 2 --emulate-world 8 --emulate-rank 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 n8-rank3', d['ms_per_step'])"
}
for r in 1 2 3; do
  one default ""
  for v in "$@"; do one $v $GRAFT_REPO_ROOT/profiles/variants/libbhray_$v.so; done
done
