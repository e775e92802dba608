#!/bin/bash
# GPU box, repo root: renders every rank's row tiles of an N-way partition in turn on ONE GPU (bench.py --emulate-world N --emulate-rank r;
# before the gather) with the driver's 20-frame blocks and with long blocks - 1920x1080 at N = 2, 4, 8 (balanced slabs; N = 8 also with the
# interleaved stripes of rounds 1-3), configs[3] 3840x2160 and configs[4] 7680x4320 / 5 levels / 2048 iterations at N = 8.  The slowest rank bounds
# the multi-GPU frame rate.  Output: gpurun_out/emulate/<tag>_emulate_{1080p,4k,8k}.json (copied to profiles/).  usage: emulate_all_ranks.sh <tag>
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/emulate
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0"
run() { name=$1; shift; timeout 900 python bench.py $B "$@" > $OUT/$name.json 2>> $OUT/err.txt || echo "FAILED $name" >> $OUT/err.txt; }
emu() {  # tag, world, long-steps, extra args...
  tag=$1; N=$2; LONG=$3; shift 3
  for ((r=0; r<N; r++)); do
    run ${tag}_n${N}_r${r}_short "$@" --steps 20 --warmup 5 --emulate-world $N --emulate-rank $r
    run ${tag}_n${N}_r${r}_long "$@" --steps $LONG --warmup 40 --min-seconds 0.3 --emulate-world $N --emulate-rank $r
  done
}
G4="--width 3840 --height 2160"; G8="--width 7680 --height 4320 --levels 5 --max-iterations 2048"
run 1080p_n1_short --steps 20 --warmup 5;  run 1080p_n1_long --steps 2000 --warmup 100 --min-seconds 0.3


emu 1080p 8 2000; emu 1080p 4 1000; emu 1080p 2 1000
emu 1080pstripes 8 2000 --partition stripes


python - "$TAG" <<'PY'
import json, os, sys
tag = sys.argv[1]
d = "gpurun_out/emulate"
def load(n):
    try: return json.load(open(os.path.join(d, n)))
    except Exception: return None
for res_name, base, groups in (("1080p", "1080p", [("1080p", 8), ("1080p", 4), ("1080p", 2), ("1080pstripes", 8)]), ("4k", "4k", [("4k", 8), ("4kstripes", 8)]), ("8k", "8k", [("8k", 8), ("8kstripes", 8)])):
    out = {"note": "one GPU renders ONE rank's row tiles of an N-way partition, before the gather (bench.py --emulate-world N --emulate-rank r); scaling = N = 1 ms per frame / slowest "
                   "rank's ms per frame; redundancy = sum of the ranks' integrator steps / the undivided frame's; valu_frac = THIS RANK's algorithmic flops / its time / the VALU peak"}
    for kind in ("short", "long"):
        n1 = load(f"{base}_n1_{kind}.json")
        if not n1: continue
        out[f"n1_{kind}"] = {"ms_per_step": n1["ms_per_step"], "mrays_per_s": n1["value"], "steps": n1["steps"], "workload": n1["config"]["workload"], "ladder": n1["config"]["ladder"],
                              "valu_frac": n1["valu"]["frac"], "integrator_steps_per_frame": n1["counters"]["steps"]}
        for t, N in groups:
            rows = []
            for r in range(N):
                j = load(f"{t}_n{N}_r{r}_{kind}.json")
                if j: rows.append({"rank": r, "ms_per_step": j["ms_per_step"], "valu_frac": j["valu"]["frac"], "frames_per_batch": j["config"]["frames_per_batch"], "steps": j["steps"],
                                   "integrator_steps_per_frame": round(j["valu"]["algorithmic_flops_per_frame"] / 390.0)})
            if not rows: continue
            j0 = load(f"{t}_n{N}_r0_{kind}.json")
            worst = max(x["ms_per_step"] for x in rows); mean = sum(x["ms_per_step"] for x in rows) / len(rows)
            out[f"{t}_n{N}_{kind}"] = {"partition": j0["config"]["partition"], "slowest_ms_per_step": worst, "mean_ms_per_step": round(mean, 5), "scaling": round(n1["ms_per_step"] / worst, 3),
                                       "scaling_if_perfectly_balanced": round(n1["ms_per_step"] / mean, 3),
                                       "redundancy": round(sum(x["integrator_steps_per_frame"] for x in rows) / float(n1["counters"]["steps"]), 4) if len(rows) == N else None,
                                       "ranks": rows}
    json.dump(out, open(os.path.join(d, f"{tag}_emulate_{res_name}.json"), "w"), indent=1)
    print(res_name, {k: (v["scaling"], v["slowest_ms_per_step"], v["redundancy"]) for k, v in out.items() if isinstance(v, dict) and "ranks" in v})
PY
tail -3 $OUT/err.txt
