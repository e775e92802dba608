#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root.  Collects, for the default bench command:
#   1. rocprofv3 --kernel-trace --stats                      (per-kernel durations)
#   2. PMC pass: FETCH_SIZE            } separate passes, kernel-trace only — never combined with
#   3. PMC pass: WRITE_SIZE            } sys/hip/hsa traces (MI355X_MICROARCH.md §rocprofv3 PMC slots)
#   4. PMC pass: SQ instruction/cycle counters
# Outputs go to gpurun_out/prof_<tag>/ ; profiles/summarize.py turns them into profiles/<tag>_*.{csv,json}.
TAG=${1:-r02}
ARGS=${2:-"--steps 96 --warmup 24 --no-cpu-baseline"}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f1 -o bench -- python $ROOT/bench.py $ARGS --frames-in-flight 1 > $OUT/stats_f1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o bench -- python $ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -o bench -- python $ROOT/bench.py $ARGS > $OUT/sq2.log 2>&1
# the same command once more, un-profiled, with the PMC summary of THIS session: its JSON line carries valu.issue
python $ROOT/profiles/summarize.py $TAG $OUT/summary > $OUT/summarize.log 2>&1
python $ROOT/bench.py $ARGS --pmc-json $OUT/summary/pmc_traffic.json > $OUT/final.log 2>&1
grep -h '^{' $OUT/final.log $OUT/stats.log $OUT/stats_f1.log | head -5 > $OUT/bench_lines.jsonl
ls -R $OUT | head -40
