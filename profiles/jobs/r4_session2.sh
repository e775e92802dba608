#!/bin/bash
# GPU box, repo root.  Round 4, session 2: the GPU suite after the prune + slab partition, then per-rank emulation with balanced slabs
# (1080p N = 2, 4, 8; 4K and 8K N = 8) at the driver's 20-frame blocks and with long blocks.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/s2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc=$?" ; tail -5 gpurun_out/s2/pytest.log
OUT=$ROOT/gpurun_out/s2/emu
mkdir -p $OUT
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0"
run() { name=$1; shift; timeout 600 python bench.py $B "$@" > $OUT/$name.json 2>> $OUT/err.txt || echo "FAILED $name" >> $OUT/err.txt; }
emu() {  # tag, world, long-steps, geometry args...
  tag=$1; N=$2; LONG=$3; shift 3
  for ((r=0; r<N; r++)); do
    run ${tag}_n${N}_r${r}_short "$@" --steps 20 --warmup 5 --emulate-world $N --emulate-rank $r
    run ${tag}_n${N}_r${r}_long "$@" --steps $LONG --warmup 10 --min-seconds 0.3 --emulate-world $N --emulate-rank $r
  done
}
run 1080p_n1_short --steps 20 --warmup 5
run 1080p_n1_long --steps 400 --warmup 32
emu 1080p 8 400
emu 1080p 4 400
emu 1080p 2 400
emu 4k 8 200 --width 3840 --height 2160
emu 8k 8 60 --width 7680 --height 4320 --levels 5 --max-iterations 2048
# the old partition again in the same session, slowest ranks only (7 at 1080p, 5 at 4K), for an A/B on one box
run 1080p_stripes_n8_r7_long --steps 400 --warmup 10 --min-seconds 0.3 --emulate-world 8 --emulate-rank 7 --partition stripes
run 1080p_stripes_n8_r3_short --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 3 --partition stripes
# the real N>1 path on one GPU (tiles travel as RCCL send/recv-to-self), balanced slabs, verified against the whole frame
timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 5 > gpurun_out/s2/b_8dup.json 2> gpurun_out/s2/b_8dup.err; echo "8dup rc=$?"
python - <<'PY'
import json, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
d = os.path.join(root, "gpurun_out", "s2", "emu")
def load(n):
    try: return json.load(open(os.path.join(d, n)))
    except Exception: return None
res = {}
n1 = {k: load(f"1080p_n1_{k}.json") for k in ("short", "long")}
for tag, N in (("1080p", 8), ("1080p", 4), ("1080p", 2), ("4k", 8), ("8k", 8)):
    for kind in ("short", "long"):
        rows = []
        for r in range(N):
            j = load(f"{tag}_n{N}_r{r}_{kind}.json")
            if j: rows.append({"rank": r, "ms_per_step": j["ms_per_step"], "valu_frac": j["valu"]["frac"], "frames_per_batch": j["config"]["frames_per_batch"], "steps": j["steps"],
                               "slab_row0": j["config"]["partition"]["slab_row0"]})
        if rows:
            worst = max(x["ms_per_step"] for x in rows); mean = sum(x["ms_per_step"] for x in rows) / len(rows)
            res[f"{tag}_n{N}_{kind}"] = {"slowest_ms_per_step": worst, "mean_ms_per_step": round(mean, 5), "ranks": rows}
res["n1"] = {k: {"ms_per_step": v["ms_per_step"], "mrays_per_s": v["value"]} for k, v in n1.items() if v}
json.dump(res, open(os.path.join(root, "gpurun_out", "s2", "emulate_balanced.json"), "w"), indent=1)
for k, v in res.items():
    if k != "n1": print(k, v["slowest_ms_per_step"], v["mean_ms_per_step"], [x["ms_per_step"] for x in v["ranks"]])
print(res["n1"])
PY
tail -5 $OUT/err.txt; tail -3 gpurun_out/s2/b_8dup.err; python -c "
import json; d=json.load(open('gpurun_out/s2/b_8dup.json')); print(d['value'], d['ms_per_step'], d['config']['verified_frames'], d['config']['partition'], d['gather'])"
