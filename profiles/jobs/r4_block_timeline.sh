#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/tl; cd /tmp && export TMPDIR=/tmp
for mode in ladder temporal; do
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/prof_$mode -o t -- python $R/profiles/jobs/r4_block_timeline.py $mode 2>&1 | grep "block ms"
python - <<P
import csv, glob
f = glob.glob("$R/gpurun_out/tl/prof_$mode/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "bhray" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = [[rows[0]]]
for b in rows[1:]:
    if int(b["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in groups[-1]) > 10_000_000: groups.append([])
    groups[-1].append(b)
g = groups[-1]
t0 = int(g[0]["Start_Timestamp"])
out = open("$R/gpurun_out/tl/r04_${TL_TAG:-rank3}_block_timeline_$mode.txt", "w")
for r in g:
    line = "%8.1f %8.1f  %7.1f us  grid %7s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Kernel_Name"][:58])
    print(line); out.write(line + "\n")
print("$mode kernels", len(g), "block span us", (max(int(x["End_Timestamp"]) for x in g) - t0) / 1e3)
P
rm -rf $R/gpurun_out/tl/prof_$mode
done
