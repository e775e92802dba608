cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full; O=gpurun_out/full
export GPU_MAX_HW_QUEUES=64
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|Traceback" | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], 'issue', d['host_issue_ms_per_step'], 'sustained', d['sustained']['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['secondary']['frac'], 'cpu', d['cpu_baseline']['value'])"
