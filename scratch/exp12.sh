run() { python bench.py --steps 192 --warmup 32 --no-cpu-baseline --min-seconds 0.2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for f in 8 12 16 24 32; do echo "N=8 fpb $f: rank0 $(run --emulate-world 8 --emulate-rank 0 --frames-per-batch $f) rank7 $(run --emulate-world 8 --emulate-rank 7 --frames-per-batch $f)"; done
for f in 4 8 12 16; do echo "N=4 fpb $f: rank0 $(run --emulate-world 4 --emulate-rank 0 --frames-per-batch $f) rank3 $(run --emulate-world 4 --emulate-rank 3 --frames-per-batch $f)"; done
for f in 2 4 8; do echo "N=2 fpb $f: rank0 $(run --emulate-world 2 --emulate-rank 0 --frames-per-batch $f) rank1 $(run --emulate-world 2 --emulate-rank 1 --frames-per-batch $f)"; done
for f in 1 2 4; do echo "N=1 fpb $f: $(run --frames-per-batch $f)"; done
echo "N=8 fpb16 fif 12: $(run --emulate-world 8 --emulate-rank 0 --frames-per-batch 16 --frames-in-flight 12)"
