#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0 --no-partition-feedback"
run() { name=$1; shift; timeout 600 python bench.py $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sequence']; print('$name', 'static', d['ms_per_step'], 'ladder', s['ladder'], 'temporal', s['temporal'], s['verified_frames'])"; }
run n1_orbit --steps 20 --warmup 5 --sequence orbit
run n1_time --steps 20 --warmup 5 --sequence time
run n8r5_orbit --steps 20 --warmup 5 --sequence orbit --emulate-world 8 --emulate-rank 5
run n8r5_time --steps 20 --warmup 5 --sequence time --emulate-world 8 --emulate-rank 5
run 4k_n1_orbit --width 3840 --height 2160 --steps 20 --warmup 5 --sequence orbit
run 4k_n8r5_orbit --width 3840 --height 2160 --steps 20 --warmup 5 --sequence orbit --emulate-world 8 --emulate-rank 5
run 4k_n8r5_time --width 3840 --height 2160 --steps 20 --warmup 5 --sequence time --emulate-world 8 --emulate-rank 5
