#!/bin/bash
# pair march diagnosis: one frame in flight (the last-level launch alone on the device), slot occupancy, VALU counters
mkdir -p gpurun_out/pair
R=$GRAFT_REPO_ROOT
for lib in libbhray.so libbhray_pair.so; do
  BHRAY_TRACE_DENSE=1 BHRAY_LIB=$R/bhusie_amd/$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sequence none --no-extra-legs --frames-in-flight 1 --sustained-steps 0 > gpurun_out/pair/f1_$lib.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/pair/f1_$lib.json').read().strip().splitlines()[-1]); print('$lib one frame in flight, dense forced: ms/frame', d['ms_per_step'])"
  BHRAY_TRACE_DENSE=1 BHRAY_LIB=$R/bhusie_amd/$lib python - <<P
import bhusie_amd as B, numpy as np
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
rp = B.RayPass(cfg, device=0, counters=True, speculative_levels=2)
rp.set_textures(*T.textures(small=False)); rp.set_uniforms(*T.uniforms(integration_method=1, max_iterations=2000)); rp.render(); rp.read_hdr()
c = rp.counters(); s = rp.scheduling_counters()
print("$lib", "steps", c["steps"], "wave_steps", s["wave_steps"], "slots per wave-step", c["steps"] / s["wave_steps"])
P
done
cd /tmp && export TMPDIR=/tmp
for lib in libbhray.so libbhray_pair.so; do
BHRAY_LIB=$R/bhusie_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pair/prof_$lib -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sequence none --no-extra-legs --sustained-steps 0 > /dev/null 2>&1
python - <<P
import csv, glob
for f in glob.glob("$R/gpurun_out/pair/prof_$lib/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("$lib", r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
for f in glob.glob("$R/gpurun_out/pair/prof_$lib/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "trace" in r["Kernel_Name"]]
    if rows:
        r = rows[len(rows)//2]
        print({k: r[k] for k in r if any(t in k for t in ("Grid", "Workgroup", "LDS", "Scratch", "VGPR", "SGPR"))})
P
BHRAY_LIB=$R/bhusie_amd/$lib timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pair/pmc_$lib -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sequence none --no-extra-legs --sustained-steps 0 > /dev/null 2>&1
python - <<P
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/pair/pmc_$lib/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "trace" in r["Kernel_Name"]:
            d[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in d.items():
        print("$lib", k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
P
rm -rf $R/gpurun_out/pair/pmc_$lib $R/gpurun_out/pair/prof_$lib
done
