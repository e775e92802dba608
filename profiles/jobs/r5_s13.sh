cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s13; O=gpurun_out/s13
export GPU_MAX_HW_QUEUES=64
timeout 1500 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_slabs.py tests/test_gpu_repartition.py tests/test_gpu_handoff.py tests/test_gpu_multirank.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
BHRAY_ISSUE_THREADS=0 timeout 900 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_repartition.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
timeout 600 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 20 --no-extra-legs --no-cpu-baseline --min-seconds 1.5 > $O/p8_packed.json 2> $O/p8_packed.err; echo rc=$?
python -c "
import json; d=json.load(open('$O/p8_packed.json')); p=d['config']['partition']
print('p8 packed', d['value'], d['ms_per_step'], 'issue', d['host_issue_ms_per_step'], 'sustained', d['sustained'] and d['sustained']['ms_per_step'], 'verified', d['config']['verified_frames']); print(p['mode'], p['slab_row0'], p['calibration_ms'], p['imbalance_first_last']); print(d['gather']); print(d['expected_scaling'])"
timeout 600 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 20 --no-extra-legs --no-cpu-baseline --min-seconds 1.5 --gather-sky > $O/p8_sky.json 2> $O/p8_sky.err; echo rc=$?
python -c "
import json; d=json.load(open('$O/p8_sky.json'))
print('p8 sky', d['value'], d['ms_per_step'], 'sustained', d['sustained'] and d['sustained']['ms_per_step'], 'verified', d['config']['verified_frames']); print(d['gather']['receive_ms_per_batch'], d['gather']['deinterleave_ms_per_batch'], d['gather']['bytes_received_per_frame'])"
