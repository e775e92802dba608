import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
rp = B.RayPass(cfg, frames_in_flight=22, speculative_levels=2)
rp.set_textures(*tex); rp.set_uniforms(*u)
def spin(dt):
    t = time.perf_counter()
    while time.perf_counter() - t < dt: pass
for extra in (0.0, 20e-6, 40e-6, 80e-6):
    blocks, issue = [], []
    for rep in range(40):
        rp.sync(); t0 = time.perf_counter()
        for _ in range(20):
            rp.render()
            if extra: spin(extra)
        t1 = time.perf_counter(); rp.sync(); t2 = time.perf_counter()
        blocks.append(t2 - t0); issue.append(t1 - t0)
    blocks.sort(); issue.sort()
    print("extra host delay per frame %3.0f us: issue of 20 frames %.3f ms, block median %.3f ms (min %.3f)" % (extra * 1e6, issue[20] * 1e3, blocks[20] * 1e3, blocks[0] * 1e3), flush=True)
