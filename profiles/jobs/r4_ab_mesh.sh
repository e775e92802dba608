#!/bin/bash
# GPU box, repo root.  A/B of mesh-kernel builds on the mesh workload (configs[2]): driver-style 20-frame blocks, 200-frame blocks, and one
# frame at a time (frames in flight 1).  usage: r4_ab_mesh.sh <rounds> <variant>...   (variants: profiles/variants/libbhray_<v>.so; "default" = the shipped library)
cd $GRAFT_REPO_ROOT
rounds=$1; shift
one() {
  L=$2; [ "$1" = default ] && L=""
  for cfg in "--steps 20 --warmup 5" "--steps 200 --warmup 20" "--steps 20 --warmup 5 --frames-in-flight 1"; do
    BHRAY_LIB=$L timeout 300 python bench.py $cfg --workload mesh --no-extra-legs --no-cpu-baseline --sustained-steps 0 --min-seconds 1.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$cfg'.replace('--steps ','s').replace(' --warmup ','w').replace(' --frames-in-flight 1',' F1'), d['value'], d['ms_per_step'])"
  done
}
for ((r=0; r<rounds; r++)); do for v in "$@"; do one $v $GRAFT_REPO_ROOT/profiles/variants/libbhray_$v.so; done; done
