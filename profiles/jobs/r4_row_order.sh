#!/bin/bash
# the level rows nearest the row the hole projects to are classified first (their rays - the longest - enter the queue first): suite + numbers
mkdir -p gpurun_out/ro
timeout 240 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -1
timeout 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_fused.py 2>&1 | grep -E "passed|failed|rror" | tail -2
B="--no-cpu-baseline --no-extra-legs --sequence none --steps 20 --warmup 5"
for t in "n1:--sustained-steps 200" "r0:--sustained-steps 0 --emulate-world 8 --emulate-rank 0" "r3:--sustained-steps 0 --emulate-world 8 --emulate-rank 3" "q1:--sustained-steps 0 --emulate-world 4 --emulate-rank 1" "k4:--sustained-steps 0 --width 3840 --height 2160 --emulate-world 8 --emulate-rank 4" "f1:--sustained-steps 0 --frames-in-flight 1" "m1:--sustained-steps 0 --workload mesh" "mf:--sustained-steps 0 --workload mesh --frames-in-flight 1"; do
    n=${t%%:*}; a=${t#*:}
    timeout 300 python bench.py $B $a > gpurun_out/ro/${n}.json 2>/dev/null
done
python -c "
import json
g=lambda n: json.loads(open('gpurun_out/ro/%s.json' % n).read().strip().splitlines()[-1])
print('N=1', g('n1')['value'], g('n1')['sustained']['mrays_per_s'], 'one frame', g('f1')['ms_per_step'], 'mesh', g('m1')['value'], g('mf')['ms_per_step'], '|', {n: g(n)['ms_per_step'] for n in ('r0','r3','q1','k4')})"
cd bhusie_amd && timeout 120 ./bhray_render --dropin 60 --rk | tail -1
