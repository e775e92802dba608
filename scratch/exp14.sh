run() { python bench.py --steps 100 --warmup 32 --no-cpu-baseline --min-seconds 0.2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['latency_ms_one_frame_in_flight_by_mode'])"; }
echo "base: $(run)"
for v in maxilp iterilp maxocc; do echo "$v: $(BHRAY_LIB=$PWD/scratch/variants/libbhray_$v.so run)"; done
