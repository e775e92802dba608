#!/bin/bash
cd $GRAFT_REPO_ROOT
bash profiles/emulate_all_ranks.sh r04
# RCCL kernels beside the trace kernels: 8 partitions on one GPU (tiles travel as send/recv-to-self), the driver's block length, balanced slabs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_rccl -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-legs --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 5 > $R/gpurun_out/r4_rccl.log 2>&1
cp $(find $R/gpurun_out/r4_rccl -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r04_kernel_stats_8partitions_one_gpu.csv 2>/dev/null
grep -h '^{' $R/gpurun_out/r4_rccl.log | head -1 > $R/gpurun_out/r04_bench_8partitions_one_gpu_profiled.json
rm -rf $R/gpurun_out/r4_rccl
head -8 $R/gpurun_out/r04_kernel_stats_8partitions_one_gpu.csv | cut -c1-200
cd $R; python bench.py --no-cpu-baseline --no-extra-legs --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 5 > gpurun_out/r04_bench_8partitions_one_gpu.json 2> gpurun_out/r04_bench_8partitions_one_gpu.err; echo rc=$?
