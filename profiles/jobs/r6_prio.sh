#!/bin/bash
# R6.5: wave priority by predicted ray length - (1) one frame at a time, the latency build with several class bands; (2) the dense builds too (-DBHRAY_WAVE_PRIO=3):
# the driver's 20-frame blocks and 400-frame blocks, RK / Euler / mesh, alternating with the shipped library.
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/r6_prio; mkdir -p $OUT
V=profiles/variants
python profiles/jobs/r6_lat_ab.py --quick bhusie_amd/libbhray.so $V/libbhray_p1.so $V/libbhray_pA.so $V/libbhray_pB.so $V/libbhray_pC.so $V/libbhray_pD.so 2 > $OUT/lat_bands.txt 2>&1
run() { # label lib args
  env BHRAY_LIB=$PWD/$2 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --sustained-steps 0 --warmup 5 --min-seconds 1.5 $3 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 |$3|', d['value'], d['ms_per_step'])" >> $OUT/blocks.txt
}
for rnd in 1 2; do for a in "--steps 20" "--steps 400" "--steps 20 --integrator euler" "--steps 400 --integrator euler" "--steps 20 --workload mesh" "--steps 400 --workload mesh"; do
  run base bhusie_amd/libbhray.so "$a"; run p3 $V/libbhray_p3.so "$a"
done; done
