#!/bin/bash
# GPU box, repo root.  Round 4, VERDICT item 1b/1d: per-rank emulation of an 8-way row partition for configs[3] (3840x2160) and
# configs[4] (7680x4320, 5 levels, 2048 iterations) at the driver's 20-frame blocks and at long blocks, and a rocprofv3 kernel trace of
# an 8-partition run on one GPU (tiles travel as RCCL send/recv-to-self) that shows the RCCL kernels beside the trace kernels.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4_emu
mkdir -p $OUT
cd $ROOT
B="--no-cpu-baseline --no-extra-legs"
run() { name=$1; shift; timeout 600 python bench.py $B "$@" > $OUT/$name.json 2>> $OUT/err.txt || echo "FAILED $name" >> $OUT/err.txt; }
for cfgname in 4k 8k; do
  if [ $cfgname = 4k ]; then G="--width 3840 --height 2160"; LONG=200; else G="--width 7680 --height 4320 --levels 5 --max-iterations 2048"; LONG=60; fi
  run ${cfgname}_n1_short $G --steps 20 --warmup 5
  run ${cfgname}_n1_long $G --steps $LONG --warmup 10 --min-seconds 0.3
  for r in 0 1 2 3 4 5 6 7; do
    run ${cfgname}_n8_r${r}_short $G --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r
    run ${cfgname}_n8_r${r}_long $G --steps $LONG --warmup 10 --min-seconds 0.3 --emulate-world 8 --emulate-rank $r
  done
done
python - <<'PY'
import json, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
d = os.path.join(root, "gpurun_out", "r4_emu")
def load(n):
    try: return json.load(open(os.path.join(d, n)))
    except Exception as e: return None
for c in ("4k", "8k"):
    out = {"note": "one GPU renders ONE rank's row tiles of an 8-way partition (27-row interleaved stripes, frames per batch auto), before the gather; "
                   "scaling = N=1 ms per frame / slowest rank's ms per frame"}
    for kind in ("short", "long"):
        n1 = load(f"{c}_n1_{kind}.json")
        if not n1: continue
        out[f"n1_{kind}"] = {"ms_per_step": n1["ms_per_step"], "mrays_per_s": n1["value"], "steps": n1["steps"], "workload": n1["config"]["workload"], "ladder": n1["config"]["ladder"]}
        rows = []
        for r in range(8):
            j = load(f"{c}_n8_r{r}_{kind}.json")
            if j: rows.append({"rank": r, "ms_per_step": j["ms_per_step"], "valu_frac": j["valu"]["frac"], "frames_per_batch": j["config"]["frames_per_batch"], "steps": j["steps"]})
        if rows:
            worst = max(x["ms_per_step"] for x in rows)
            mean = sum(x["ms_per_step"] for x in rows) / len(rows)
            out[f"n8_{kind}"] = {"ranks": rows, "slowest_ms_per_step": worst, "mean_ms_per_step": round(mean, 5), "scaling": round(n1["ms_per_step"] / worst, 3),
                                 "scaling_if_balanced": round(n1["ms_per_step"] / mean, 3)}
    json.dump(out, open(os.path.join(root, "gpurun_out", f"r04_emulate_{c}.json"), "w"), indent=1)
    print(c, {k: (v.get("scaling"), v.get("slowest_ms_per_step")) for k, v in out.items() if k.startswith("n8")})
PY
# RCCL kernels beside the trace kernels: 8 partitions on one GPU, the driver's block length
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4_rccl -o bench -- python $ROOT/bench.py $B --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 5 > $ROOT/gpurun_out/r4_rccl.log 2>&1
cp $(find $ROOT/gpurun_out/r4_rccl -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/r04_kernel_stats_8partitions_one_gpu.csv 2>/dev/null
grep -h '^{' $ROOT/gpurun_out/r4_rccl.log | head -1 > $ROOT/gpurun_out/r04_bench_8partitions_one_gpu_profiled.json
rm -rf $ROOT/gpurun_out/r4_rccl
head -12 $ROOT/gpurun_out/r04_kernel_stats_8partitions_one_gpu.csv
tail -5 $OUT/err.txt
