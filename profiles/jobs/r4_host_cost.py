import time, bhusie_amd as B, ctypes as C
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
L = B.lib()
for kw in (dict(speculative_levels=2), dict(speculative_levels=3), dict(temporal=True)):
    rp = B.RayPass(cfg, device=0, frames_in_flight=22, **kw)
    rp.set_textures(*T.textures(small=False)); u = T.uniforms(integration_method=1); rp.set_uniforms(*u)
    for _ in range(44): rp.render()
    rp.sync()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20): L.bhray_render(rp._h)
        t1 = time.perf_counter(); rp.sync()
        best = min(best, (t1 - t0) / 20)
    t0 = time.perf_counter()
    for _ in range(1000): L.bhray_set_uniforms(rp._h, u[0], u[1], u[2])
    su = (time.perf_counter() - t0) / 1000
    print(kw, "bhray_render host time per call: %.1f us; set_uniforms %.1f us" % (best * 1e6, su * 1e6))
    rp.close()
