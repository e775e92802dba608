import sys, os, time, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
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
print(os.environ.get("BHRAY_LIB","default").split("/")[-1], os.environ.get("BHRAY_TRACE_BLOCKS_PER_CU"), "fused_S2", lat(speculative_levels=2, fused=True), flush=True)
