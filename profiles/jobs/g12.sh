cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g12
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/g12/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/g12/pytest.log
grep -E "passed|failed|rc=|^FAILED|Error" gpurun_out/g12/pytest.log | head -20
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/g12/bench_driver.json 2> gpurun_out/g12/bench_driver.err
python -c "
import json; d=json.load(open('gpurun_out/g12/bench_driver.json')); print(d['value'], d['ms_per_step'], d['handoff'], d['roofline']['avg_launch_ms'])"
tail -n 3 gpurun_out/g12/bench_driver.err
