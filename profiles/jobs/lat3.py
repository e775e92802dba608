import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cam, bh = B.Camera(), B.BlackHole()
det = B.RayDetails(integration_method=1, step_size=0.15, max_iterations=2000, angle_division_threshold=0.02, time=0.0)
def lat(cfg, n=25, **kw):
    rp = B.RayPass(cfg, device=0, frames_in_flight=1, **kw)
    rp.set_textures(*tex); rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
    rp.close()
    ts = sorted(ts[5:])
    return round(ts[len(ts) // 2] * 1e3, 4)
full = B.ladder_for_frame((1920, 1080), 3, 4)
l0 = B.ladder_from_base((73, 41), 3, 1)
out = {"lib": os.environ.get("BHRAY_LIB", "default").split("/")[-1], "level0_alone": lat(l0), "S2": lat(full, speculative_levels=2), "S3": lat(full, speculative_levels=3),
       "temporal_static": lat(full, temporal=True)}
print(json.dumps(out), flush=True)
