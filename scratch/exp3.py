import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
for kw in (dict(frames_in_flight=16, speculative_levels=2), dict(frames_in_flight=1, speculative_levels=2)):
    rp = B.RayPass(cfg, counters=True, **kw)
    rp.set_textures(*tex)
    rp.set_uniforms(B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
    rp.render(); rp.sync()
    print(kw, rp.scheduling_counters(), [ (rp.level_counters(l)['steps'], rp.level_counters(l)['traced']) for l in range(4)])
    rp.close()
