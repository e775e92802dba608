#!/bin/bash
# the moving scene (bench.py --sequence orbit: time += 1/60 and a strafing camera, every frame its own uniforms) per RANK of an 8-way
# partition, emulated on one GPU before the gather: the ladder and BHRAY_F_TEMPORAL, 20-frame blocks and one frame at a time.
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/seqranks; rm -rf $OUT; mkdir -p $OUT; cd $ROOT
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0 --sequence orbit --steps 20 --warmup 5"
timeout 600 python bench.py $B > $OUT/n1.json 2>> $OUT/err.txt
for r in 0 1 2 3 4 5 6 7; do
  timeout 600 python bench.py $B --emulate-world 8 --emulate-rank $r > $OUT/r$r.json 2>> $OUT/err.txt || echo "FAILED $r" >> $OUT/err.txt
done
python - <<'PY'
import json
d = "gpurun_out/seqranks"
def seq(n):
    j = json.loads(open(f"{d}/{n}.json").read().strip().splitlines()[-1]); return j["sequence"], j
n1, j1 = seq("n1")
out = {"note": "bench.py --sequence orbit --steps 20 --warmup 5; one GPU renders ONE rank's rows of an 8-way partition (balanced slabs from a calibration frame at the sequence's first "
               "camera), before the gather; scaling = N = 1 ms per frame / the slowest rank's; every run verified its temporal frames byte for byte against the ladder",
       "n1": {"ladder": n1["ladder"], "temporal": n1["temporal"], "verified_frames": n1["verified_frames"]}, "ranks": []}
for r in range(8):
    s, j = seq(f"r{r}")
    out["ranks"].append({"rank": r, "ladder": s["ladder"], "temporal": s["temporal"], "verified_frames": s["verified_frames"], "frames_per_batch": s["frames_per_batch"], "static_scene_ms_per_step": j["ms_per_step"]})
for mode in ("ladder", "temporal"):
    for key in ("ms_per_step", "latency_ms_one_frame_in_flight"):
        worst = max(x[mode][key] for x in out["ranks"])
        out[f"{mode}_{key}_slowest_rank"] = worst
        out[f"{mode}_{key}_scaling"] = round(n1[mode][key] / worst, 3)
json.dump(out, open(f"{d}/r04_sequence_ranks.json", "w"), indent=1)
print({k: v for k, v in out.items() if k not in ("ranks", "note")})
PY
tail -3 $OUT/err.txt
