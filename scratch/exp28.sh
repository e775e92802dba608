#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
python - <<'PY'
import bhusie_amd as B
rp = B.RayPass(B.ladder_from_base((24, 14), 3, 2))
print("selftest", rp.selftest())
PY
python scratch/exp24.py
timeout 600 python bench.py --no-cpu-baseline
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5
} > gpurun_out/exp28.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_literal.py tests/test_gpu_temporal.py -x -q > gpurun_out/exp28_pytest.log 2>&1
