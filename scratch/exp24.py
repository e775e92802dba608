import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_from_base((73, 41), 3, 1)
u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
rp = B.RayPass(cfg, frames_in_flight=1, timing=True, counters=False)
rp.set_textures(*tex); rp.set_uniforms(*u)
for _ in range(3): rp.render()
rp.sync(); rp.timing()
for _ in range(20): rp.render(); rp.sync()
tm = rp.timing()
print(os.environ.get("BHRAY_LIB", "base").split("_")[-1], "level-0 only: trace %.4f ms classify %.4f ms per frame" % (tm.trace_ms / tm.frames, tm.classify_ms / tm.frames))
