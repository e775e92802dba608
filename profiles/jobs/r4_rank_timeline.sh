#!/bin/bash
# timeline of ONE 20-frame block of rank 3 of an 8-way partition (1080p): every kernel's start / end relative to the block's first kernel
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/tl; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/prof -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --sustained-steps 0 --sequence none --emulate-world 8 --emulate-rank 3 --min-seconds 0.05 > /dev/null 2>&1
python - <<P
import csv, glob
f = glob.glob("$R/gpurun_out/tl/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "bhray" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = rows[-56:]                      # the last kernels of the run: the last timed block(s); blocks show as gaps (host sync between them)
t0 = int(g[0]["Start_Timestamp"])
out = open("$R/gpurun_out/tl/r04_rank3_block_timeline.txt", "w")
for r in g:
    line = "%8.1f %8.1f  %7.1f us  grid %7s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Kernel_Name"][:60])
    print(line); out.write(line + "\n")
print("kernels", len(g), "block span us", (max(int(x["End_Timestamp"]) for x in g) - t0) / 1e3)
P
rm -rf $R/gpurun_out/tl/prof
