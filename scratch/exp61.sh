#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build+smoke ok')" > gpurun_out/final2_smoke.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final2_pytest.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final2_bench.json 2> gpurun_out/final2_bench.err
tail -1 gpurun_out/final2_smoke.log; grep -E "passed|failed" gpurun_out/final2_pytest.log; python -c "import json; d=json.load(open('gpurun_out/final2_bench.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline']['value'])"
