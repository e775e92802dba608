#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_superset.py tests/test_gpu_temporal.py tests/test_gpu_bvh_stack.py tests/test_gpu_fused.py -q > gpurun_out/s3/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s3/pytest.log
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0"
run() { name=$1; shift; timeout 600 python bench.py $B "$@" 2>> gpurun_out/s3/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['config']['frames_per_batch'])"; }
K4="--width 3840 --height 2160 --steps 20 --warmup 5 --emulate-world 8"
echo "== 4K, 20-frame blocks, rank 4 (balanced) / rank 5 (stripes): frames per batch, trace build"
for f in 2 3 4 5 7 10; do run 4k_bal_r4_fpb$f $K4 --emulate-rank 4 --frames-per-batch $f; done
for f in 4 5 7 10; do run 4k_str_r5_fpb$f $K4 --emulate-rank 5 --frames-per-batch $f --partition stripes; done
for f in 4 7; do BHRAY_TRACE_DENSE=1 run 4k_bal_r4_dense_fpb$f $K4 --emulate-rank 4 --frames-per-batch $f; BHRAY_TRACE_DENSE=0 run 4k_bal_r4_latency_fpb$f $K4 --emulate-rank 4 --frames-per-batch $f; done
for s in 3; do run 4k_bal_r4_spec3 $K4 --emulate-rank 4 --speculative-levels 3; done
echo "== 1080p N=8, 2000-frame blocks"
L="--steps 2000 --warmup 100 --min-seconds 0.3 --emulate-world 8"
for r in 0 4 5; do run 1080p_bal_r${r}_2000 $L --emulate-rank $r; done
run 1080p_str_r7_2000 $L --emulate-rank 7 --partition stripes
run 1080p_str_r5_2000 $L --emulate-rank 5 --partition stripes
python bench.py $B --steps 2000 --warmup 100 --min-seconds 0.3 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1_2000', d['value'], d['ms_per_step'])"
echo "== drop-in (C++)"
GPU_MAX_HW_QUEUES=8 timeout 300 ./bhusie_amd/bhray_render --dropin 60 --rk
echo "== mesh variants"
bash profiles/jobs/r4_ab_mesh.sh 1 default base inl5 inl4 inl5_16_8 inl4_16_8 inl4_1_1 inl5_1_1 inl4_32_16
tail -5 gpurun_out/s3/err.txt
