#!/bin/bash
# GPU box, repo root: renders every rank's row tiles of an N-way partition in turn on ONE GPU (bench.py --emulate-world N --emulate-rank r;
# before the gather), for N = 2, 4, 8, with the driver's 20-frame blocks and with long blocks.  The slowest rank bounds the
# multi-GPU frame rate.  Output: gpurun_out/emulate_all_ranks.json (copied to profiles/<tag>_emulate_all_ranks.json).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/emulate
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/n1_short.json 2>/dev/null
python bench.py --steps 400 --warmup 32 --no-cpu-baseline --no-extra-legs > $OUT/n1_long.json 2>/dev/null
for N in 2 4 8; do for ((r=0; r<N; r++)); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --emulate-world $N --emulate-rank $r > $OUT/n${N}_r${r}_short.json 2>/dev/null
  python bench.py --steps 400 --warmup 32 --no-cpu-baseline --no-extra-legs --emulate-world $N --emulate-rank $r --min-seconds 0.2 > $OUT/n${N}_r${r}_long.json 2>/dev/null
done; done
python - <<'PY'
import json, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
d = os.path.join(root, "gpurun_out", "emulate")
def load(n): return json.load(open(os.path.join(d, n)))
out = {"note": "one GPU renders ONE rank's row tiles of an N-way partition (27-row interleaved stripes, frames per batch auto), before the gather; "
               "scaling = N=1 ms per frame / slowest rank's ms per frame; valu.frac = THIS RANK's algorithmic flops / its time / the VALU peak",
       "n1": {k: {"ms_per_step": load(f"n1_{k}.json")["ms_per_step"], "mrays_per_s": load(f"n1_{k}.json")["value"]} for k in ("short", "long")}}
for N in (2, 4, 8):
    for kind in ("short", "long"):
        rows = []
        for r in range(N):
            j = load(f"n{N}_r{r}_{kind}.json")
            rows.append({"rank": r, "ms_per_step": j["ms_per_step"], "valu_frac": j["valu"]["frac"], "frames_per_batch": j["config"]["frames_per_batch"]})
        worst = max(x["ms_per_step"] for x in rows)
        out[f"n{N}_{kind}"] = {"ranks": rows, "slowest_ms_per_step": worst, "scaling": round(out["n1"][kind]["ms_per_step"] / worst, 3),
                               "blocks": "20-frame blocks (--steps 20 --warmup 5)" if kind == "short" else "400-frame blocks"}
json.dump(out, open(os.path.join(root, "gpurun_out", "emulate_all_ranks.json"), "w"), indent=1)
print(json.dumps({k: (v["scaling"], v["slowest_ms_per_step"]) for k, v in out.items() if k[0] == "n" and k[1].isdigit() and "_" in k}))
PY
rm -rf $OUT
