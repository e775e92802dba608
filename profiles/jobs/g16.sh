cd $GRAFT_REPO_ROOT
bash profiles/collect.sh r03 > gpurun_out/collect_r03.log 2>&1
tail -n 12 gpurun_out/collect_r03.log | cut -c1-300
bash profiles/emulate_all_ranks.sh > gpurun_out/emulate_r03.log 2>&1
tail -n 2 gpurun_out/emulate_r03.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_r03.json 2> gpurun_out/bench_driver_r03.err
python -c "
import json; d=json.load(open('gpurun_out/bench_driver_r03.json')); print(d['value'], d['ms_per_step'], json.dumps(d['handoff'])[:700]); print(d['latency_ms_one_frame_in_flight_by_mode']); print(d['literal']['mrays_per_s'], d['literal']['eval_fma']['mrays_per_s'])"
