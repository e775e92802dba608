#!/bin/bash
ROOT=/root/repo; mkdir -p $ROOT/gpurun_out/trace20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/trace20 -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --min-seconds 0.05 --no-cpu-baseline > $ROOT/gpurun_out/trace20/log.txt 2>&1
ls -R $ROOT/gpurun_out/trace20 | head
