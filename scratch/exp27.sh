#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
BHRAY_LIB=scratch/variants/libbhray_prof.so python scratch/exp24.py > gpurun_out/exp27.log 2>&1
