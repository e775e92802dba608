run() { python bench.py --steps 200 --warmup 32 --no-cpu-baseline --min-seconds 0.3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for t in 0 12 20; do echo "T=$t: $(BHRAY_LIB=$PWD/bhusie_amd/libbhray_t$t.so run) $(BHRAY_LIB=$PWD/bhusie_amd/libbhray_t$t.so run)"; done
echo "T=32: $(run) $(run)"
