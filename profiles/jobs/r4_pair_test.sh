#!/bin/bash
# pair march: byte-identity tests, A/B against the default library (20-frame blocks + sustained), kernel trace of the pair library
mkdir -p gpurun_out/pair
timeout 1200 python -m pytest tests/test_gpu_pair.py -x -q -m gpu > gpurun_out/pair/test.log 2>&1; echo "pair tests rc=$?"
tail -5 gpurun_out/pair/test.log
for round in 1 2; do
for lib in libbhray.so libbhray_pair.so; do
  BHRAY_LIB=$PWD/bhusie_amd/$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sequence none --no-extra-legs > gpurun_out/pair/bench_$lib.$round.json 2> gpurun_out/pair/bench_$lib.err; echo "$lib rc=$?"
  python - <<P
import json
d=json.loads(open("gpurun_out/pair/bench_$lib.$round.json").read().strip().splitlines()[-1])
print("$lib", d["value"], d["ms_per_step"], (d.get("sustained") or {}).get("value"), d.get("scheduling"), d.get("roofline",{}).get("kernel_ms"))
P
done
done
cd /tmp && export TMPDIR=/tmp
for lib in libbhray.so libbhray_pair.so; do
BHRAY_LIB=$GRAFT_REPO_ROOT/bhusie_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pair/prof_$lib -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sequence none --no-extra-legs --sustained-steps 0 > /dev/null 2>&1
python - <<P
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pair/prof_$lib/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("$lib", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pair/prof_$lib/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "trace" in r["Kernel_Name"]]
    if rows:
        r = rows[len(rows)//2]
        print({k: r[k] for k in r if k in ("Kernel_Name","Grid_Size_X","Workgroup_Size_X","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count")})
P
done
