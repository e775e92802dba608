python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --steps 256 --warmup 48 > /tmp/e.json; python -c "
import json
d=json.loads(open('/tmp/e.json').read().strip().splitlines()[-1]); print('disk', d['value'], d['ms_per_step'], 'ms', d['roofline']['isolated']['level_trace_ms'])"; done
python bench.py --no-cpu-baseline --steps 192 --warmup 32 --workload mesh > /tmp/e.json; python -c "
import json
d=json.loads(open('/tmp/e.json').read().strip().splitlines()[-1]); print('mesh', d['value'], d['ms_per_step'], 'ms')"
