#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
BHRAY_TRACE_DENSE=1 BHRAY_LIB=scratch/variants/libbhray_powstat.so python scratch/exp46.py > gpurun_out/exp46.log 2>&1
