#!/bin/bash
mkdir -p gpurun_out/tb
timeout 240 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -p no:cacheprovider > gpurun_out/tb/fused.log 2>&1; echo "fused rc=$?"
grep -E "passed|failed|rror" gpurun_out/tb/fused.log | tail -2
timeout 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_fused.py > gpurun_out/tb/suite.log 2>&1; echo "rest rc=$?"
grep -E "passed|failed|rror" gpurun_out/tb/suite.log | tail -2
