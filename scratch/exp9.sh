run() { python bench.py --steps 100 --warmup 32 --no-cpu-baseline --min-seconds 0.3 --workload mesh "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['latency_ms_one_frame_in_flight'], d['roofline']['isolated']['level_trace_ms'])"; }
echo "base: $(run)"; echo "base: $(run)"
for v in top512 stk4 stk8 top512stk8; do echo "$v: $(BHRAY_LIB=$PWD/bhusie_amd/libbhray_$v.so run)"; done
BHRAY_LIB=$PWD/bhusie_amd/libbhray_top512stk8.so python -m pytest tests/test_gpu_parity.py -q -k "mesh" 2>&1 | tail -2
