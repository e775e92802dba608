#!/bin/bash
# BHRAY_F_TEMPORAL with frames in flight after the grid change (its launches follow the ctx's grid rule when there are several frame slots):
# temporal tests, the moving sequences on one GPU and for every rank of an 8-way partition
mkdir -p gpurun_out/tg
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0 --steps 20 --warmup 5 $FEEDBACK"
for seq in time orbit; do
  timeout 600 python bench.py $B --sequence $seq > gpurun_out/tg/n1_$seq.json 2>/dev/null
  for r in 0 1 2 3 4 5 6 7; do timeout 600 python bench.py $B --sequence $seq --emulate-world 8 --emulate-rank $r > gpurun_out/tg/r${r}_$seq.json 2>/dev/null; done
done
python - <<'P'
import json
d = "gpurun_out/tg"
def seq(n):
    j = json.loads(open(f"{d}/{n}.json").read().strip().splitlines()[-1]); return j["sequence"], j
out = {"note": "bench.py --sequence {time,orbit} --steps 20 --warmup 5 on the final tree (the temporal mode's launches follow the ctx's grid rule when frames are in flight); one GPU renders ONE rank's rows of an "
               "8-way partition (balanced slabs from a calibration frame at the sequence's first camera, two feedback rounds), before the gather; scaling = N = 1 LADDER ms per frame / the slowest rank's; "
               "every run verified its temporal frames byte for byte against the ladder"}
for kind in ("time", "orbit"):
    n1, j1 = seq(f"n1_{kind}")
    o = {"n1": {"ladder": n1["ladder"], "temporal": n1["temporal"], "verified_frames": n1["verified_frames"]}, "ranks": []}
    for r in range(8):
        s, j = seq(f"r{r}_{kind}")
        o["ranks"].append({"rank": r, "ladder": s["ladder"], "temporal": s["temporal"], "verified_frames": s["verified_frames"], "frames_per_batch": s["frames_per_batch"]})
    for mode in ("ladder", "temporal"):
        for key in ("ms_per_step", "latency_ms_one_frame_in_flight"):
            worst = max(x[mode][key] for x in o["ranks"])
            o[f"{mode}_{key}_slowest_rank"] = worst
            o[f"{mode}_{key}_scaling_against_the_n1_ladder"] = round(n1["ladder"][key] / worst, 3)
    out[kind] = o
    print(kind, {k: v for k, v in o.items() if k not in ("ranks",)})
import os
json.dump(out, open(f"{d}/r04_sequence_ranks%s.json" % ("_temporal_bounds" if os.environ.get("FEEDBACK") else ""), "w"), indent=1)
P
