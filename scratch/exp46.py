import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
for method in (1,):
    u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=method).uniform())
    rp = B.RayPass(cfg, frames_in_flight=1, counters=True)
    rp.set_textures(*tex); rp.set_uniforms(*u); rp.render(); rp.sync()
    for l in range(4):
        c = rp.level_counters(l)
        lib = B.lib() if hasattr(B, "lib") else None
        from bhusie_amd.layouts import BhrayCounters
        import ctypes as C
        from bhusie_amd._lib import lib as L
        cc = BhrayCounters(); L().bhray_get_level_counters(rp._h, l, C.byref(cc))
        s = cc.scheduling()
        print("level", l, "steps", c["steps"], "wave_steps", s.get("wave_steps"), "lanes that took the power", s.get("rays_adopted"), "wave-steps with the power", c["node_pairs"],
              "-> lanes %.5f of lane-steps, wave-steps %.4f" % (s.get("rays_adopted") / max(1, c["steps"]), c["node_pairs"] / max(1, s.get("wave_steps"))),
              "| disk hits", c["disk_hits"], "shade-phase invocations", c["triangles"], "-> %.2f lanes per invocation" % (c["disk_hits"] / max(1, c["triangles"])))
