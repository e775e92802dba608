#!/bin/bash
# N = 1, the driver's 20-frame blocks: frames per batch, speculative / superset levels, frames in flight
mkdir -p gpurun_out/n1
run() { tag=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sequence none --no-extra-legs --sustained-steps 0 "$@" > gpurun_out/n1/$tag.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/n1/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])"; }
run base
run fpb2 --frames-per-batch 2
run fpb4 --frames-per-batch 4
run fpb5 --frames-per-batch 5
run fpb10 --frames-per-batch 10
run s3 --speculative-levels 3
run u1 --superset-levels 1
run u2 --superset-levels 2
run fif20 --frames-in-flight 20
run fif32 --frames-in-flight 32
run base2
