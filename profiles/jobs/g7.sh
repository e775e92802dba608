cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g7
run() { name=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --emulate-world 8 --emulate-rank 0 "$@" > gpurun_out/g7/$name.json 2>> gpurun_out/g7/err.txt; python -c "
import json; d=json.load(open('gpurun_out/g7/$name.json')); print('$name', d['value'], d['ms_per_step'], d['timed_blocks']['block_ms'])"; }
run s2_b10 --frames-per-batch 10
run s3_b10 --frames-per-batch 10 --speculative-levels 3
run s3_b20 --frames-per-batch 20 --speculative-levels 3
run s3_b5 --frames-per-batch 5 --speculative-levels 3
run s2u2_b10 --frames-per-batch 10 --superset-levels 2
run s3_b10_f2 --frames-per-batch 10 --speculative-levels 3 --frames-in-flight 2
run s3_b7 --frames-per-batch 7 --speculative-levels 3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/g7/n1.json 2>> gpurun_out/g7/err.txt; python -c "
import json; d=json.load(open('gpurun_out/g7/n1.json')); print('n1', d['value'], d['ms_per_step'])"
