cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s25; O=gpurun_out/s25
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/tests.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
timeout 600 python bench.py --workload mesh 2>/dev/null | tail -1 > $O/bench_mesh.json
python - <<'P'
import json
for n in ("default", "mesh"):
    d = json.loads(open("gpurun_out/s25/bench_%s.json" % n).read())
    print(n, d["value"], d["ms_per_step"], d["roofline"])
P
