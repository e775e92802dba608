# round 5, session 2: why a 20-frame block of the 8-partition ctx on ONE GPU takes 3-4x the sum of its rank probes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2; O=gpurun_out/s2
export GPU_MAX_HW_QUEUES=64
B="python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 20 --no-extra-legs --no-cpu-baseline --frames-in-flight 6 --partition-feedback-rounds 1 --min-seconds 1.5"
for grid in 0 256 128 64 32; do
  if [ $grid = 0 ]; then unset BHRAY_TRACE_GRID; else export BHRAY_TRACE_GRID=$grid; fi
  timeout 600 $B > $O/p8_grid$grid.json 2> $O/p8_grid$grid.err; echo rc=$?
  python -c "
import json; d=json.load(open('$O/p8_grid$grid.json')); p=d['config']['partition']
print('p8 grid $grid', d['value'], d['ms_per_step'], 'blocks', d['timed_blocks']['block_ms'], 'issue', d['host_issue_ms_per_step'], 'sustained', d['sustained'] and d['sustained']['ms_per_step'], 'sumprobes', round(sum(p['probe_ms_per_frame'] or [0]),4))"
done
unset BHRAY_TRACE_GRID
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p8 -- $B --sustained-steps 0 --no-partition-feedback > $O/p8_profiled.json 2> $O/p8_profiled.err; echo rc=$?
f=$(find $O/prof -name '*kernel_trace.csv' | head -1); ls -la $f
python profiles/jobs/r5_timeline.py $f $O/p8_timeline.json
rm -rf $O/prof
