import os, sys, time, math; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
bh, det = B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform()
def path(da, n=16):
    out = []
    for i in range(n):
        a = da * i
        pos = (19.0 * math.sin(a), 15.0 * da * i, -19.0 * math.cos(a))
        nn = math.sqrt(sum(v * v for v in pos))
        out.append((B.Camera(position=pos, forward=tuple(-v / nn for v in pos)).uniform(), bh, det))
    return out
for da in (0.0, 0.0005, 0.002, 0.005, 0.02):
    res = {}
    for name, kw in (("S2", dict(speculative_levels=2)), ("temporal", dict(temporal=True))):
        rp = B.RayPass(cfg, frames_in_flight=1, counters=False, **kw)
        rp.set_textures(*tex)
        us = path(da)
        for u in us[:3]:
            rp.set_uniforms(*u); rp.render(); rp.sync()
        ts = []
        for u in us[3:]:
            rp.set_uniforms(*u)
            t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
        res[name] = round(sorted(ts)[len(ts) // 2] * 1e3, 3)
        rp.close()
    rp = B.RayPass(cfg, frames_in_flight=1, counters=True, temporal=True)
    rp.set_textures(*tex)
    tr = []
    for u in path(da)[:6]:
        rp.set_uniforms(*u); rp.render(); tr.append(rp.counters()["traced"])
    rp.close()
    print("orbit step %.4f rad/frame (~%.1f px):" % (da, da / (1.0 / 1080)), res, "traced per frame (temporal)", tr)
