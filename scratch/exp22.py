import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
for timing in (True, False, True, False):
    rp = B.RayPass(cfg, frames_in_flight=20, speculative_levels=2, timing=timing)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    for _ in range(40): rp.render()
    rp.sync()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(400): rp.render()
        rp.sync()
        res.append(1920 * 1080 * 400 / (time.perf_counter() - t0) / 1e6)
    print("timing events", timing, [round(x) for x in res])
    rp.close()
