run() { python bench.py --steps 200 --warmup 32 --no-cpu-baseline --min-seconds 0.3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['latency_ms_one_frame_in_flight'], d['roofline']['isolated']['level_trace_ms'])"; }
echo "base: $(run)"; echo "base: $(run)"
for m in 4 8 16 32; do echo "refill min $m: $(BHRAY_LIB=$PWD/scratch/variants/libbhray_rf$m.so run)"; done
