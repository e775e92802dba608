run() { python bench.py --steps 200 --warmup 32 --no-cpu-baseline --min-seconds 0.3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['latency_ms_one_frame_in_flight'])"; }
for v in t0 w128 w64; do echo "$v: $(BHRAY_LIB=$PWD/bhusie_amd/libbhray_$v.so run) | $(BHRAY_LIB=$PWD/bhusie_amd/libbhray_$v.so run)"; done
for v in w64; do for g in 256 384 768; do echo "$v grid $g: $(BHRAY_TRACE_GRID=$g BHRAY_LIB=$PWD/bhusie_amd/libbhray_$v.so run)"; done; done
