# round 5, session 1: baseline on this round's box, host issue of the 8-partition ctx (one thread), the 8-partition one-GPU gap
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s1; O=gpurun_out/s1
export GPU_MAX_HW_QUEUES=64
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], 'issue', d['host_issue_ms_per_step'], 'sustained', d['sustained']['ms_per_step'])"
timeout 300 python profiles/jobs/r5_host_issue.py $O/host_issue_one_thread.json 2>&1 | tail -5
for fif in 22 6 3; do
  timeout 600 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline --frames-in-flight $fif --partition-feedback-rounds 1 > $O/p8_fif$fif.json 2> $O/p8_fif$fif.err; echo rc=$?
  python -c "
import json; d=json.load(open('$O/p8_fif$fif.json')); p=d['config']['partition']
print('p8 fif $fif', d['value'], d['ms_per_step'], 'issue', d['host_issue_ms_per_step'], 'sustained', d['sustained'] and d['sustained']['ms_per_step'], 'probes', p['probe_ms_per_frame'], 'sum', sum(p['probe_ms_per_frame'] or [0]), 'fpb', d['config']['frames_per_batch'], 'gather', d['gather']['receive_ms_per_batch'], d['gather']['deinterleave_ms_per_batch'])"
done
