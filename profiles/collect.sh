#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root.  Collects, for the driver's bench command and three more workloads:
#   1. rocprofv3 --kernel-trace --stats                      (per-kernel durations; the bench line of the SAME profiled run is kept
#      beside it: rocprofv3 changes how much the 22 frames in flight overlap, so only that line is comparable with the CSV)
#   2. PMC pass: FETCH_SIZE            } separate passes, kernel-trace only - never combined with
#   3. PMC pass: WRITE_SIZE            } sys/hip/hsa traces (MI355X_MICROARCH.md, rocprofv3 PMC slots)
#   4. PMC passes: SQ instruction / cycle counters
# Outputs go to gpurun_out/prof_<tag>/<workload>/ ; profiles/summarize.py turns them into profiles/<tag>_*.{csv,json}.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
BASE="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs"
cd /tmp && export TMPDIR=/tmp
prof() {   # workload-name, extra bench args
  W=$1; shift
  OUT=$ROOT/gpurun_out/prof_$TAG/$W
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py $BASE "$@" > $OUT/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python $ROOT/bench.py $BASE "$@" > $OUT/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python $ROOT/bench.py $BASE "$@" > $OUT/write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -o bench -- python $ROOT/bench.py $BASE "$@" > $OUT/sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o bench -- python $ROOT/bench.py $BASE "$@" > $OUT/sq.log 2>&1
  grep -h '^{' $OUT/stats.log | head -1 > $OUT/bench_line_profiled.json
  python $ROOT/bench.py $BASE "$@" > $OUT/bench_line.json 2> $OUT/bench_line.err     # the same command, un-profiled
}
prof default
OUT=$ROOT/gpurun_out/prof_$TAG/default
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f1 -o bench -- python $ROOT/bench.py $BASE --frames-in-flight 1 > $OUT/stats_f1.log 2>&1
prof euler --integrator euler
prof mesh --workload mesh
prof 4k --width 3840 --height 2160
python $ROOT/profiles/summarize.py $TAG $ROOT/gpurun_out/prof_$TAG/summary > $ROOT/gpurun_out/prof_$TAG/summarize.log 2>&1
# drop the raw traces (large); the summaries stay
for W in default euler mesh 4k; do for d in stats fetch write sq sq2 stats_f1; do rm -rf $ROOT/gpurun_out/prof_$TAG/$W/$d; done; done
tail -n 30 $ROOT/gpurun_out/prof_$TAG/summarize.log
