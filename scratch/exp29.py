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
for name, da, dy in (("static", 0.0, 0.0), ("slow 0.002 rad + 0.03/frame", 0.002, 0.03), ("bench path 0.02 rad + 0.3/frame", 0.02, 0.3)):
    P = path(16, da, dy)
    for margin, radius in ((1.0, "0"), (0.9, "1,4"), (0.8, "1,4"), (0.8, "0,4"), (0.8, "2,4"), (0.7, "1,4"), (0.7, "2,4")):
        os.environ["BHRAY_TEMPORAL_MARGIN"] = str(margin); os.environ["BHRAY_TEMPORAL_RADIUS"] = radius
        rp = B.RayPass(cfg, frames_in_flight=1, temporal=True)
        rp.set_textures(*tex)
        ts, traced = [], []
        for i, u in enumerate(P):
            rp.set_uniforms(*u)
            t0 = time.perf_counter(); rp.render(); rp.sync(); dt = time.perf_counter() - t0
            if i >= 3: ts.append(dt); traced.append(0)
        rp.close()
        print("%-34s margin %.2f radius %s: median %.3f ms  min %.3f  traced/frame %d" % (name, margin, radius, sorted(ts)[len(ts) // 2] * 1e3, min(ts) * 1e3, sum(traced) // len(traced)), flush=True)
