cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g13
timeout 900 python -m pytest tests -m gpu -q -k "mesh or config2 or handoff" --timeout 600 2>&1 | grep -E "passed|failed|^FAILED" | head
for v in default inl8 inl6; do
  if [ $v = default ]; then L=$GRAFT_REPO_ROOT/bhusie_amd/libbhray.so; else L=$GRAFT_REPO_ROOT/profiles/variants/libbhray_$v.so; fi
  BHRAY_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --workload mesh > gpurun_out/g13/mesh_$v.json 2>> gpurun_out/g13/err.txt
  BHRAY_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --workload mesh --frames-in-flight 1 > gpurun_out/g13/mesh_f1_$v.json 2>> gpurun_out/g13/err.txt
  python -c "
import json; a=json.load(open('gpurun_out/g13/mesh_$v.json')); b=json.load(open('gpurun_out/g13/mesh_f1_$v.json')); print('$v', a['value'], a['ms_per_step'], 'one frame', b['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/g13/$c -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --workload mesh > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/g13/$c/**/*counter_collection.csv", recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "trace_kernel<1, true, false" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items(): print(k, "KB per launch", sum(v)/len(v), "launches", len(v))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/g13/$c
done
