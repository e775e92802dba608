#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python scratch/exp29.py > gpurun_out/exp29.log 2>&1
