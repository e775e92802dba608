cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s14; O=gpurun_out/s14
export GPU_MAX_HW_QUEUES=64
timeout 1500 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_slabs.py tests/test_gpu_repartition.py tests/test_gpu_handoff.py tests/test_gpu_multirank.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for i in 1 2; do
timeout 600 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 20 --no-extra-legs --no-cpu-baseline --min-seconds 1.5 > $O/p8_packed$i.json 2> $O/p8_packed$i.err; echo rc=$?
python -c "
import json; d=json.load(open('$O/p8_packed$i.json')); p=d['config']['partition']; g=d['gather']
print('p8 packed', d['value'], d['ms_per_step'], d['timed_blocks']['block_ms'], 'issue', d['host_issue_ms_per_step'], 'sustained', d['sustained'] and d['sustained']['ms_per_step'], 'verified', d['config']['verified_frames'], 'recv', g['receive_ms_per_batch'], g['deinterleave_ms_per_batch'])"
done
