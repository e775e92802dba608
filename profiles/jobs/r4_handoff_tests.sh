#!/bin/bash
# the hand-off tests added in round 4 (stale sky image, refused async read, the shim's forms from the C++ program), then the whole GPU suite
mkdir -p gpurun_out/h
timeout 900 python -m pytest tests/test_gpu_handoff.py -x -q -m gpu > gpurun_out/h/handoff.log 2>&1; echo "handoff rc=$?"
tail -15 gpurun_out/h/handoff.log
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/h/suite.log 2>&1; echo "suite rc=$?"
tail -5 gpurun_out/h/suite.log
