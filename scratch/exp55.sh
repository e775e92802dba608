#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
BHRAY_TRACE_DENSE=1 BHRAY_LIB=scratch/variants/libbhray_phasestat.so python scratch/exp55.py > gpurun_out/exp55.log 2>&1
