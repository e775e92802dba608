cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -q --timeout 300 > gpurun_out/g5/pytest_fused.log 2>&1
tail -n 4 gpurun_out/g5/pytest_fused.log | head -3
cat > /tmp/lat.py <<'PY'
import sys, os, time, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cam, bh = B.Camera(), B.BlackHole()
det = B.RayDetails(integration_method=1, step_size=0.15, max_iterations=2000, angle_division_threshold=0.02, time=0.0)
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
def lat(**kw):
    rp = B.RayPass(cfg, device=0, frames_in_flight=1, **kw)
    rp.set_textures(*tex); rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
    ts = []
    for i in range(15):
        t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
    rp.close()
    return round(sorted(ts[3:])[6] * 1e3, 4)
out = {}
for name, kw in (("plain_S2", dict(speculative_levels=2)), ("fused_S2", dict(speculative_levels=2, fused=True)), ("fused_S0", dict(fused=True)), ("fused_S3", dict(speculative_levels=3, fused=True))):
    out[name] = lat(**kw)
    print(name, out[name], flush=True)
json.dump(out, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/g5/lat.json"), "w"))
PY
timeout 300 python /tmp/lat.py
for bpc in 1 2; do echo "bpc $bpc"; BHRAY_TRACE_BLOCKS_PER_CU=$bpc timeout 300 python /tmp/lat.py 2>&1 | grep fused; done
for fpb in 10 20; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --emulate-world 8 --emulate-rank 0 --frames-per-batch $fpb --fused > gpurun_out/g5/emu8_fused_b$fpb.json 2>> gpurun_out/g5/emu.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/g5/emu*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
