#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_edge_cases.py -x -q -k fuzz_every > gpurun_out/exp56.log 2>&1
