#!/bin/bash
# Evidence for design decisions that were MEASURED AND REJECTED in round 2 (run on the GPU box via gpurun, from the repo root).
# The variants are compile-time flags of the same sources, built into profiles/variants/ (make ... EXTRA=-D...):
#   libbhray_stk8.so       -DBHRAY_BVH_LDS_STACK=8     first 8 entries of the BVH traversal stack in LDS
#   libbhray_top512.so     -DBHRAY_BVH_LDS_TOP=512     top 512 BVH nodes (levels 0-8 of the breadth-first order) staged in LDS
#   libbhray_mailbox32.so  -DBHRAY_MAILBOX_T=32        drain merging through the per-block LDS mailbox
# Output: gpurun_out/prof_exp/<variant>/{stats,fetch}/... and gpurun_out/prof_exp/bench_lines.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_exp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MESH="--workload mesh --steps 64 --warmup 16 --no-cpu-baseline --min-seconds 0.2"
DISK="--steps 96 --warmup 24 --no-cpu-baseline --min-seconds 0.2"
for v in base stk8 top512; do
  LIB=$ROOT/bhusie_amd/libbhray.so; [ $v != base ] && LIB=$ROOT/profiles/variants/libbhray_$v.so
  BHRAY_LIB=$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mesh_$v/stats -o bench -- python $ROOT/bench.py $MESH > $OUT/mesh_$v.stats.log 2>&1
  BHRAY_LIB=$LIB rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/mesh_$v/fetch -o bench -- python $ROOT/bench.py $MESH > $OUT/mesh_$v.fetch.log 2>&1
  echo "mesh $v: $(grep -h '^{' $OUT/mesh_$v.stats.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame, one frame in flight', d['latency_ms_one_frame_in_flight'], 'ms')")" >> $OUT/bench_lines.txt
done
for v in base mailbox32; do
  LIB=$ROOT/bhusie_amd/libbhray.so; [ $v != base ] && LIB=$ROOT/profiles/variants/libbhray_$v.so
  BHRAY_LIB=$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/disk_$v/stats -o bench -- python $ROOT/bench.py $DISK > $OUT/disk_$v.stats.log 2>&1
  BHRAY_LIB=$LIB rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/disk_$v/sq -o bench -- python $ROOT/bench.py $DISK > $OUT/disk_$v.sq.log 2>&1
  for i in 1 2 3; do echo "disk $v (un-profiled run $i): $(BHRAY_LIB=$LIB python $ROOT/bench.py --steps 200 --warmup 32 --no-cpu-baseline --min-seconds 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'Mrays/s')")" >> $OUT/bench_lines.txt; done
done
cat $OUT/bench_lines.txt
