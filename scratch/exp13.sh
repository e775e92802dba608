run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for f in 2 4 5 7 10 20; do echo "N=8 20 steps fpb $f: rank0 $(run --emulate-world 8 --emulate-rank 0 --frames-per-batch $f) fif8: $(run --emulate-world 8 --emulate-rank 0 --frames-per-batch $f --frames-in-flight 8)"; done
for f in 1 2 4 5 10; do echo "N=4 20 steps fpb $f: rank3 $(run --emulate-world 4 --emulate-rank 3 --frames-per-batch $f)"; done
for f in 1 2 4 5; do echo "N=2 20 steps fpb $f: rank0 $(run --emulate-world 2 --emulate-rank 0 --frames-per-batch $f)"; done
echo "N=1 20 steps: $(run)"
