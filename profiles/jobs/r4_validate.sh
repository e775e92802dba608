#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/v1/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/v1/pytest.log | tail -1; grep "^FAILED" gpurun_out/v1/pytest.log | head
python -c "import __graft_entry__ as g; g.smoke()"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/v1/bench_default.json 2> gpurun_out/v1/bench_default.err; echo "bench rc=$?"; tail -2 gpurun_out/v1/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --static-time --no-extra-legs --no-cpu-baseline > gpurun_out/v1/bench_static.json 2>/dev/null
timeout 600 python bench.py --gpus 4 --devices 0,0,0,0 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/v1/bench_4dup.json 2> gpurun_out/v1/bench_4dup.err; echo "4dup rc=$?"; tail -2 gpurun_out/v1/bench_4dup.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/v1/bench_default.json')); s=json.load(open('gpurun_out/v1/bench_static.json')); q=json.load(open('gpurun_out/v1/bench_4dup.json'))
print('default', d['value'], d['ms_per_step'], 'sustained', d['sustained']['mrays_per_s'], 'static-time', s['value'], s['sustained']['mrays_per_s'])
print('seq', d['sequence']['ladder'], d['sequence']['temporal'], d['sequence']['verified_frames'])
print('dropin', {k: v['mrays_per_s'] for k, v in d['dropin']['legs'].items()})
print('4dup', q['value'], q['config']['verified_frames'], q['config']['partition']['slab_row0'], q['config']['partition']['probe_ms_per_frame'])
PY
