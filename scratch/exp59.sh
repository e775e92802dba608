#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for d in 0 1; do
  BHRAY_TRACE_DENSE=$d timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_latency_build_and_dense_build_deliver_the_same_frame --deselect tests/test_gpu_edge_cases.py::test_fuzz_every_build_and_mode_delivers_the_same_frame > gpurun_out/exp59_dense$d.log 2>&1
  echo "dense=$d: $(grep -E 'passed|failed' gpurun_out/exp59_dense$d.log | tail -1)"
done
