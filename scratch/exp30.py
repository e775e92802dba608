import os, sys, time, math; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
bh, det = B.BlackHole(), B.RayDetails(integration_method=1)
def path(n, da, dy):
    out = []
    for i in range(n):
        a = da * i
        pos = (19.0 * math.sin(a), dy * i, -19.0 * math.cos(a))
        nn = math.sqrt(sum(v * v for v in pos))
        out.append((B.Camera(position=pos, forward=tuple(-v / nn for v in pos)).uniform(), bh.uniform(), det.uniform()))
    return out
for name, da, dy in (("slow", 0.002, 0.03), ("bench", 0.02, 0.3)):
    P = path(9, da, dy)
    for margin, radius in ((1.0, "0"), (0.7, "0,2"), (0.7, "1,2"), (0.7, "1,4"), (0.7, "2,4"), (0.5, "1,4"), (0.5, "2,4"), (0.8, "1,4"), (0.9, "1,4")):
        os.environ["BHRAY_TEMPORAL_MARGIN"] = str(margin); os.environ["BHRAY_TEMPORAL_RADIUS"] = radius
        rp = B.RayPass(cfg, frames_in_flight=1, temporal=True, counters=True, timing=True)
        rp.set_textures(*tex)
        for i, u in enumerate(P):
            rp.set_uniforms(*u)
            rp.render(); rp.sync()
            if i >= 5:
                lc = [rp.level_counters(l) for l in range(4)]
                sch = [{}, {}]
                tm = rp.timing()
                print(name, margin, radius, "frame", i, "traced per level", [c["traced"] for c in lc], "level trace ms", [round(v / max(1, tm.frames), 3) for v in list(tm.level_trace_ms)[:4]], sch[1] if sch[1] else "", flush=True)
            else:
                rp.timing()
        rp.close()
