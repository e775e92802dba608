#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python scratch/exp26.py > gpurun_out/exp26.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/exp26_pytest.log 2>&1
tail -3 gpurun_out/exp26_pytest.log
