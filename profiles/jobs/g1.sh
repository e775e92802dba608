cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
rm -f gpurun_out/literal_distance.jsonl
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -x -k "literal or handoff" > gpurun_out/g1/pytest_new.log 2>&1
echo "new rc=$?" >> gpurun_out/g1/pytest_new.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -k "not literal and not handoff" > gpurun_out/g1/pytest_rest.log 2>&1
echo "rest rc=$?" >> gpurun_out/g1/pytest_rest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/g1/bench_driver.json 2> gpurun_out/g1/bench_driver.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/g1/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/gpurun_out/g1/stats.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/g1/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/g1/kernel_stats.csv \;
rm -rf gpurun_out/g1/stats
tail -5 gpurun_out/g1/pytest_new.log gpurun_out/g1/pytest_rest.log
