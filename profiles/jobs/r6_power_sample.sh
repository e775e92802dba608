#!/bin/bash
# What clock and power does the device hold while the bench's sustained block runs?  sysfs hwmon / pp_dpm_sclk sampled at ~20 Hz beside `python bench.py`.
cd ${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/power; mkdir -p $OUT
ls /sys/class/drm/ > $OUT/drm.txt 2>&1
for d in /sys/class/drm/card*/device; do echo $d; ls $d | tr '\n' ' '; echo; ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; done >> $OUT/drm.txt 2>&1
sample() {
  while [ ! -e $OUT/stop ]; do
    for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
      echo "$(date +%s.%N) $(cat $h/freq1_input 2>/dev/null) $(cat $h/power1_average 2>/dev/null) $(cat $h/power1_input 2>/dev/null) $(cat $h/temp1_input 2>/dev/null) $(cat $h/temp2_input 2>/dev/null)"
    done
    sleep 0.05
  done
}
rm -f $OUT/stop
sample > $OUT/samples_$1.txt &
SP=$!
sleep 1
shift
"$@" > $OUT/bench.txt 2> $OUT/bench_err.txt
sleep 0.5
touch $OUT/stop; wait $SP
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^=\|^$" > $OUT/smi.txt
tail -1 $OUT/bench.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms', d['ms_per_step'], 'sustained', d.get('sustained'))"
