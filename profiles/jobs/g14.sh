cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g14
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/g14/pytest.log 2>&1
grep -E "passed|failed|^FAILED" gpurun_out/g14/pytest.log | head
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('driver-style', d['value'], d['ms_per_step'])"; done
timeout 300 python bench.py --steps 400 --warmup 32 --no-cpu-baseline --no-extra-legs | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('long', d['value'], d['ms_per_step'])"
