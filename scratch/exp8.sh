run() { python bench.py --steps 200 --warmup 32 --no-cpu-baseline --min-seconds 0.3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for i in 1 2 3; do echo "default: $(run)   no-early-out: $(BHRAY_LIB=$PWD/bhusie_amd/libbhray_ne.so run)"; done
