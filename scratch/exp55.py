import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import bhusie_amd as B
from bhusie_amd import assets
from bhusie_amd.layouts import BhrayCounters
from bhusie_amd._lib import lib as L
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
rp = B.RayPass(cfg, frames_in_flight=1, counters=True)
rp.set_textures(*tex); rp.set_uniforms(*u); rp.render(); rp.sync()
for l in range(4):
    c = rp.level_counters(l)
    cc = BhrayCounters(); L().bhray_get_level_counters(rp._h, l, C.byref(cc)); s = cc.scheduling()
    ws = s.get("wave_steps"); rounds = ws / 16.0
    print("level", l, "traced", c["traced"], "wave_steps", ws, "(~%d rounds)" % rounds, "| flat iterations", c["flat_iters"], "flat-phase invocations", c["node_pairs"], "-> %.1f lanes each" % (c["flat_iters"] / max(1, c["node_pairs"])),
          "| epilogue invocations", s.get("rays_adopted"), "-> %.1f lanes each" % (c["traced"] / max(1, s.get("rays_adopted"))), "| shade invocations", c["triangles"], "-> %.1f lanes each" % (c["disk_hits"] / max(1, c["triangles"])))
