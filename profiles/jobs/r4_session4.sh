#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/s4/emu
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/s4/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s4/pytest.log | head -3
echo "== default bench line (all legs)"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/s4/bench_default.json 2> gpurun_out/s4/bench_default.err; echo "rc=$?"; tail -2 gpurun_out/s4/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s4/bench_default.json'))
print('value', d['value'], 'sustained', d['sustained'], '\nsequence', d['sequence'], '\ndropin', d['dropin'], '\nfrac_from_profiles', d['roofline']['frac_from_profiles'], '\none_frame', d['valu']['one_frame_in_flight'])
PY
OUT=$ROOT/gpurun_out/s4/emu
B="--no-cpu-baseline --no-extra-legs --sustained-steps 0"
run() { name=$1; shift; timeout 600 python bench.py $B "$@" > $OUT/$name.json 2>> $OUT/err.txt || echo "FAILED $name" >> $OUT/err.txt; }
emu() {  # tag, world, long-steps, geometry args...
  tag=$1; N=$2; LONG=$3; shift 3
  for ((r=0; r<N; r++)); do
    run ${tag}_n${N}_r${r}_short "$@" --steps 20 --warmup 5 --emulate-world $N --emulate-rank $r
    run ${tag}_n${N}_r${r}_long "$@" --steps $LONG --warmup 40 --min-seconds 0.3 --emulate-world $N --emulate-rank $r
  done
}
run 1080p_n1_short --steps 20 --warmup 5
run 1080p_n1_long --steps 2000 --warmup 100 --min-seconds 0.3
run 4k_n1_short --width 3840 --height 2160 --steps 20 --warmup 5
run 4k_n1_long --width 3840 --height 2160 --steps 500 --warmup 20 --min-seconds 0.3
emu 1080p 8 2000
emu 4k 8 500 --width 3840 --height 2160
emu 1080p 4 1000
emu 1080p 2 1000
python - <<'PY'
import json, os
d = "gpurun_out/s4/emu"
def load(n):
    try: return json.load(open(os.path.join(d, n)))
    except Exception: return None
res = {}
for tag, N in (("1080p", 8), ("4k", 8), ("1080p", 4), ("1080p", 2)):
    for kind in ("short", "long"):
        n1 = load(f"{tag}_n1_{kind}.json")
        rows = []
        for r in range(N):
            j = load(f"{tag}_n{N}_r{r}_{kind}.json")
            if j: rows.append({"rank": r, "ms_per_step": j["ms_per_step"], "valu_frac": j["valu"]["frac"], "frames_per_batch": j["config"]["frames_per_batch"], "steps": j["steps"],
                               "slab_row0": j["config"]["partition"]["slab_row0"], "probe_ms": j["config"]["partition"]["probe_ms_per_frame"]})
        if rows and n1:
            worst = max(x["ms_per_step"] for x in rows); mean = sum(x["ms_per_step"] for x in rows) / len(rows)
            res[f"{tag}_n{N}_{kind}"] = {"n1_ms_per_step": n1["ms_per_step"], "slowest_ms_per_step": worst, "mean_ms_per_step": round(mean, 5), "scaling": round(n1["ms_per_step"] / worst, 3), "ranks": rows}
json.dump(res, open("gpurun_out/s4/emulate_balanced_feedback.json", "w"), indent=1)
for k, v in res.items(): print(k, v["scaling"], v["slowest_ms_per_step"], v["mean_ms_per_step"], [x["ms_per_step"] for x in v["ranks"]], v["ranks"][0]["slab_row0"])
PY
tail -3 $OUT/err.txt
echo "== drop-in (C++)"
GPU_MAX_HW_QUEUES=8 timeout 300 ./bhusie_amd/bhray_render --dropin 60 --rk
echo "== mesh"
bash profiles/jobs/r4_ab_mesh.sh 2 default inlc5 base
