import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
for base in ((73, 41), (217, 121)):
    cfg = B.ladder_from_base(base, 3, 1)
    for mi in (1, 25, 50, 100, 200, 400, 2000):
        u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1, max_iterations=mi).uniform())
        rp = B.RayPass(cfg, frames_in_flight=1, timing=True, counters=True)
        rp.set_textures(*tex); rp.set_uniforms(*u)
        for _ in range(3): rp.render()
        rp.sync(); rp.timing()
        for _ in range(20): rp.render(); rp.sync()
        tm = rp.timing()
        c = rp.scheduling() if hasattr(rp, "scheduling") else {}
        print(base, "max_it %4d: trace %.4f ms classify %.4f ms" % (mi, tm.trace_ms / tm.frames, tm.classify_ms / tm.frames), c)
        rp.close()
