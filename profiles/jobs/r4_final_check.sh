#!/bin/bash
mkdir -p gpurun_out/fin
timeout 240 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -1
timeout 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_fused.py 2>&1 | grep -E "passed|failed|rror" | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
B="--no-cpu-baseline --no-extra-legs --sequence none --steps 20 --warmup 5"
for t in "n1:--sustained-steps 200" "r3:--sustained-steps 0 --emulate-world 8 --emulate-rank 3" "f1:--sustained-steps 0 --frames-in-flight 1" "m1:--sustained-steps 200 --workload mesh" "mf:--sustained-steps 0 --workload mesh --frames-in-flight 1" "e1:--sustained-steps 0 --integrator euler" "k1:--sustained-steps 0 --width 3840 --height 2160"; do
    n=${t%%:*}; a=${t#*:}
    timeout 300 python bench.py $B $a > gpurun_out/fin/${n}.json 2>/dev/null
done
python -c "
import json
g=lambda n: json.loads(open('gpurun_out/fin/%s.json' % n).read().strip().splitlines()[-1])
print('N=1', g('n1')['value'], g('n1')['sustained']['mrays_per_s'], 'one frame', g('f1')['ms_per_step'], '| mesh', g('m1')['value'], g('m1')['sustained']['mrays_per_s'], g('mf')['ms_per_step'], '| euler', g('e1')['value'], '| 4K', g('k1')['value'], '| rank 3/8', g('r3')['ms_per_step'])"
cd bhusie_amd && timeout 120 ./bhray_render --dropin 60 --rk | tail -1 | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print({k: (v['mrays_per_s'], v['ms_per_frame']) for k,v in d['legs'].items()})"
