cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
# one frame at a time + the driver line with the fused latency modes
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/g4/bench_default.json 2> gpurun_out/g4/bench_default.err
# fused in the timed region, driver style
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --fused > gpurun_out/g4/bench_fused.json 2> gpurun_out/g4/bench_fused.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --fused --frames-in-flight 4 > gpurun_out/g4/bench_fused_f4.json 2>> gpurun_out/g4/bench_fused.err
# rank 0 of 8, 20-frame blocks: launch-per-level (as in round 2) against fused, several batch sizes
for fpb in 10 20; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --emulate-world 8 --emulate-rank 0 --frames-per-batch $fpb > gpurun_out/g4/emu8_plain_b$fpb.json 2>> gpurun_out/g4/emu.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --emulate-world 8 --emulate-rank 0 --frames-per-batch $fpb --fused > gpurun_out/g4/emu8_fused_b$fpb.json 2>> gpurun_out/g4/emu.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --emulate-world 8 --emulate-rank 0 --frames-per-batch 5 --fused > gpurun_out/g4/emu8_fused_b5.json 2>> gpurun_out/g4/emu.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --emulate-world 8 --emulate-rank 0 --frames-per-batch 20 --fused --frames-in-flight 2 > gpurun_out/g4/emu8_fused_b20_f2.json 2>> gpurun_out/g4/emu.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/g4/*.json")):
    try:
        d=json.load(open(f))
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("latency_ms_one_frame_in_flight_by_mode"), (d.get("handoff") or {}).get("async_pinned"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -n 5 gpurun_out/g4/*.err
