#!/bin/bash
# SQ counters of the dense trace kernels, the product library against variant builds (BHRAY_LIB), the driver's bench command.  usage: r6_pmc_ab.sh OUTDIR "bench args" lib1 lib2 ...
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/$1; ARGS=$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  BHRAY_LIB=$ROOT/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq_$n -o bench -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --sustained-steps 0 $ARGS > $O/sq_$n.log 2>&1
  BHRAY_LIB=$ROOT/$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq2_$n -o bench -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --sustained-steps 0 $ARGS > $O/sq2_$n.log 2>&1
  python - <<PY
import csv, glob, collections
for p in ("sq", "sq2"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$O/%s_$n/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "trace_kernel" not in k: continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k in tot:
        print("$n", p, k, {c: round(v / max(1, n[(k, c)])) for c, v in tot[k].items()}, "launches", max(n[(k, c)] for c in tot[k]))
PY
  rm -rf $O/sq_$n $O/sq2_$n
done
