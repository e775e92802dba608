run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['latency_ms_one_frame_in_flight'])"; }
echo "base: $(run) | $(run)"
for m in 2 4; do echo "unroll $m: $(BHRAY_LIB=$PWD/scratch/variants/libbhray_un$m.so run) | $(BHRAY_LIB=$PWD/scratch/variants/libbhray_un$m.so run)"; done
