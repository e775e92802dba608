#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0 --no-partition-feedback"
run() { name=$1; shift; timeout 600 python bench.py $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['config']['frames_per_batch'])"; }
K4="--width 3840 --height 2160 --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 5"
K1="--steps 20 --warmup 5 --emulate-world 8 --emulate-rank 5"
echo "== 4K N=8 rank 5, 20-frame blocks: uneven batches"
run 4k_default $K4
BHRAY_TRACE_DENSE=1 run 4k_default_dense $K4
BENCH_FLUSH_AT=2 run 4k_flush2_fpb6 $K4 --frames-per-batch 6
BENCH_FLUSH_AT=2,6 run 4k_flush2_6_fpb7 $K4 --frames-per-batch 7
BENCH_FLUSH_AT=1,3,7 run 4k_flush1_3_7_fpb8 $K4 --frames-per-batch 8
BENCH_FLUSH_AT=3 run 4k_flush3_fpb6 $K4 --frames-per-batch 6
BENCH_FLUSH_AT=2,5,9,14 run 4k_flush2_5_9_14 $K4 --frames-per-batch 8
BENCH_FLUSH_AT=1 run 4k_flush1_fpb5 $K4 --frames-per-batch 5
BHRAY_TRACE_DENSE=1 BENCH_FLUSH_AT=2,6 run 4k_flush2_6_fpb7_dense $K4 --frames-per-batch 7
echo "== 1080p"
run 1080p_default $K1
BENCH_FLUSH_AT=3 run 1080p_flush3 $K1 --frames-per-batch 10
BENCH_FLUSH_AT=2,6,12 run 1080p_flush2_6_12 $K1 --frames-per-batch 10
BENCH_FLUSH_AT=4,10 run 1080p_flush4_10 $K1 --frames-per-batch 10
BENCH_FLUSH_AT=5 run 1080p_flush5_fpb8 $K1 --frames-per-batch 8
