#!/bin/bash
# frames per batch again (after R4.14): 4- and 2-way 1080p partitions, 4K 8-way, and long blocks of the 8-way 1080p partition
mkdir -p gpurun_out/fp
B="--no-cpu-baseline --no-extra-legs --sequence none --warmup 5 --sustained-steps 0"
run() { tag=$1; shift; timeout 300 python bench.py $B "$@" > gpurun_out/fp/$tag.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/fp/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], 'fpb', d['config']['frames_per_batch'])"; }
for f in 4 5 10; do run n4_r1_fpb$f --steps 20 --emulate-world 4 --emulate-rank 1 --frames-per-batch $f; done
for f in 2 4 5 10; do run n2_r0_fpb$f --steps 20 --emulate-world 2 --emulate-rank 0 --frames-per-batch $f; done
for f in 2 4 5 10; do run k4_r4_fpb$f --steps 20 --width 3840 --height 2160 --emulate-world 8 --emulate-rank 4 --frames-per-batch $f; done
for f in 5 8 10 16; do run n8_r3_long_fpb$f --steps 400 --min-seconds 0.3 --emulate-world 8 --emulate-rank 3 --frames-per-batch $f; done
