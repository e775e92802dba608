#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0 --no-partition-feedback"
run() { name=$1; shift; timeout 600 python bench.py $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['config']['frames_per_batch'])"; }
K4="--width 3840 --height 2160 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 5"
K1="--steps 20 --warmup 5 --emulate-world 8 --emulate-rank 5"
echo "== 4K N=8 rank 5, 20-frame blocks: persistent blocks per CU"
run 4k_default $K4
for b in 1 2 3 4 6; do BHRAY_TRACE_BLOCKS_PER_CU=$b run 4k_bpc$b $K4; done
BHRAY_TRACE_DENSE=1 run 4k_dense $K4
BHRAY_TRACE_DENSE=0 run 4k_latency $K4
run 4k_super2 $K4 --superset-levels 2
run 4k_fif8 $K4 --frames-in-flight 8
echo "== 1080p N=8 rank 5, 20-frame blocks"
run 1080p_default $K1
for b in 1 2 3 4; do BHRAY_TRACE_BLOCKS_PER_CU=$b run 1080p_bpc$b $K1; done
BHRAY_TRACE_DENSE=1 run 1080p_dense $K1
BHRAY_TRACE_DENSE=0 run 1080p_latency $K1
for f in 4 5 10; do run 1080p_fpb$f $K1 --frames-per-batch $f; done
run 1080p_stripes $K1 --partition stripes
