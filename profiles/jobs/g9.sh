cd $GRAFT_REPO_ROOT
bash profiles/collect.sh r03 > gpurun_out/collect_r03.log 2>&1
tail -n 40 gpurun_out/collect_r03.log
bash profiles/emulate_all_ranks.sh > gpurun_out/emulate_r03.log 2>&1
tail -n 3 gpurun_out/emulate_r03.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_r03.json 2> gpurun_out/bench_driver_r03.err
