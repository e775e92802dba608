cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s7; O=gpurun_out/s7
export GPU_MAX_HW_QUEUES=64
timeout 900 python -m pytest tests/test_gpu_repartition.py tests/test_gpu_handoff.py "tests/test_gpu_edge_cases.py::test_a_batch_on_a_small_persistent_grid_traces_every_frame" -x -q -m gpu 2>&1 | tail -25
one() {  # label lib workload-args
  for cfg in "--steps 20 --warmup 5" "--steps 400 --warmup 32"; do
    BHRAY_AB_OLD_BUILD=1 BHRAY_LIB=$2 timeout 300 python bench.py $cfg $3 --no-extra-legs --no-cpu-baseline --min-seconds 2 --sustained-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d['steps'], d['value'], d['ms_per_step'])"
  done
}
for r in 1 2; do
  for wl in "" "--integrator euler" "--workload mesh"; do
    one r4 $GRAFT_REPO_ROOT/profiles/variants/libbhray_r4.so "$wl"
    one now "" "$wl"
  done
done 2>&1 | tee $O/ab_work_counter.txt
for thr in 1; do
  timeout 600 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 20 --no-extra-legs --no-cpu-baseline --min-seconds 1.5 > $O/p8_library.json 2> $O/p8_library.err; echo rc=$?
  python -c "
import json; d=json.load(open('$O/p8_library.json')); p=d['config']['partition']
print('p8 library', d['value'], d['ms_per_step'], 'issue', d['host_issue_ms_per_step'], 'sustained', d['sustained'] and d['sustained']['ms_per_step'], 'verified', d['config']['verified_frames']); print(p)"
done
tail -3 $O/p8_library.err
