#!/bin/bash
# The bench lines kept under profiles/<tag>_bench_*.json and <tag>_emulate_n*.json (run on the GPU box from the repo root).
# One JSON line per file, exactly as bench.py printed it.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/bench_$TAG
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout 900 python bench.py "$@" 2> $OUT/$name.err | grep '^{' | head -1 > $OUT/${TAG}_$name.json; echo "$name: $(python -c "import json,sys; d=json.load(open('$OUT/${TAG}_$name.json')); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" 2>&1)"; }
run bench_default
run bench_driver_style --steps 20 --warmup 5
run bench_euler --integrator euler --no-cpu-baseline
run bench_mesh --workload mesh --steps 200 --warmup 32 --no-cpu-baseline
run bench_4k --width 3840 --height 2160 --steps 100 --warmup 32 --no-cpu-baseline
run bench_8k --width 7680 --height 4320 --steps 40 --warmup 8 --no-cpu-baseline
run bench_8partitions_one_gpu --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 64 --warmup 16 --verify --no-cpu-baseline
for n in 2 4 8; do
  run emulate_n$n --emulate-world $n --emulate-rank 0 --steps 192 --warmup 32 --no-cpu-baseline
  run emulate_n${n}_driver_style --emulate-world $n --emulate-rank 0 --steps 20 --warmup 5 --no-cpu-baseline
done
