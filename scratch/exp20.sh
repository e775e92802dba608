run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['latency_ms_one_frame_in_flight'])"; }
echo "4x1: $(run) | $(run)"
for v in 2x2 4x2 2x4 4x4 8x1 8x2; do echo "$v: $(BHRAY_LIB=$PWD/scratch/variants/libbhray_c$v.so run) | $(BHRAY_LIB=$PWD/scratch/variants/libbhray_c$v.so run)"; done
BHRAY_LIB=$PWD/scratch/variants/libbhray_c4x4.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_temporal.py -x -q 2>&1 | tail -2
