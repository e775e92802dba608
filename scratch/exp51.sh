#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python scratch/exp51.py > gpurun_out/exp51.log 2>&1
