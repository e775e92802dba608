cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s4; O=gpurun_out/s4
export GPU_MAX_HW_QUEUES=64
timeout 900 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_slabs.py tests/test_gpu_handoff.py tests/test_gpu_multirank.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -15
timeout 300 python profiles/jobs/r5_host_issue.py $O/host_issue_threads.json 2>&1 | grep "^{"
BHRAY_ISSUE_THREADS=0 timeout 300 python profiles/jobs/r5_host_issue.py $O/host_issue_one_thread.json 2>&1 | grep "^{"
for thr in 1 0; do
  BHRAY_ISSUE_THREADS=$thr timeout 600 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 20 --no-extra-legs --no-cpu-baseline --partition-feedback-rounds 1 --min-seconds 1.5 > $O/p8_thr$thr.json 2> $O/p8_thr$thr.err; echo rc=$?
  python -c "
import json; d=json.load(open('$O/p8_thr$thr.json')); p=d['config']['partition']
print('p8 threads $thr', d['value'], d['ms_per_step'], 'blocks', d['timed_blocks']['block_ms'], 'issue', d['host_issue_ms_per_step'], 'sustained', d['sustained'] and d['sustained']['ms_per_step'], 'sumprobes', round(sum(p['probe_ms_per_frame'] or [0]),4), 'verified', d['config']['verified_frames'])"
done
