import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhusie_amd as B, ctypes as C
from bhusie_amd import assets, layouts
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
for method in (1, 0):
    rp = B.RayPass(cfg, counters=True, frames_in_flight=1)
    rp.set_textures(*tex)
    rp.set_uniforms(B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=method).uniform())
    rp.render(); rp.sync()
    for l in range(4):
        c = layouts.BhrayCounters()
        B.lib().bhray_get_level_counters(rp._h, l, C.byref(c))
        print("method", method, "level", l, c.as_dict()["traced"], c.as_dict()["steps"], c.scheduling())
    rp.close()
