cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/g10
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/g10/rb -o bench -- python $R/bench.py --steps 20 --warmup 5 --readback async --no-extra-legs --no-cpu-baseline > $R/gpurun_out/g10/rb.log 2>&1
cd $R
find gpurun_out/g10/rb -name "*stats*.csv" | head; for f in $(find gpurun_out/g10/rb -name "*_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-200; done
f=$(find gpurun_out/g10/rb -name "*memory_copy_trace.csv" | head -1); echo $f; head -3 $f; python - <<PY
import csv,sys
rows=list(csv.DictReader(open("$f")))
big=[r for r in rows if int(r.get("Bytes", r.get("bytes", 0)) or 0) > 30000000] if rows and ("Bytes" in rows[0] or "bytes" in rows[0]) else rows
print(len(rows), len(big), list(rows[0].keys()) if rows else None)
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in big]
if d: print("big copies: n", len(d), "avg ms", sum(d)/len(d), "min", min(d), "max", max(d))
PY
grep '^{' gpurun_out/g10/rb.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
rm -rf gpurun_out/g10/rb
