import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhusie_amd as B
from tests import common as T
tex = T.textures()
def frames(cfg, u):
    out = {}
    for name, env in (("latency", "0"), ("dense", "1")):
        os.environ["BHRAY_TRACE_DENSE"] = env
        rp = B.RayPass(cfg, counters=True, frames_in_flight=1)
        rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
        out[name] = ([rp.read_level(l) for l in range(len(cfg.sizes()))], rp.counters()); rp.close()
    return out
cfg = B.ladder_for_frame((300, 170), 3, 3)
cases = {"default": dict(), "max_it 60": dict(max_iterations=60), "max_it 150": dict(max_iterations=150), "max_it 400": dict(max_iterations=400),
         "step 0.05": dict(step_size=0.05), "step 0.3": dict(step_size=0.3), "thr 0.005": dict(angle_division_threshold=0.005), "thr 0.08": dict(angle_division_threshold=0.08),
         "time 3": dict(time=3.0), "bh off": dict(black_hole=B.BlackHole(position=(2.0, -1.0, 1.5))), "R 10": dict(black_hole=B.BlackHole(relativity_sphere_radius=10.0)),
         "R 30": dict(black_hole=B.BlackHole(relativity_sphere_radius=30.0)), "disk 2-6": dict(black_hole=B.BlackHole(accretion_disk_inner=2.0, accretion_disk_outer=6.0)),
         "cam off": dict(camera=B.Camera(position=(5.0, 3.0, -15.0), forward=(-0.3, -0.2, 0.93))), "fov 0.7": dict(camera=B.Camera(fov=0.7))}
for method in (1, 0):
    for name, kw in cases.items():
        u = T.uniforms(integration_method=method, **kw)
        f = frames(cfg, u)
        lv_same = [np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(f["latency"][0], f["dense"][0])]
        cs = f["latency"][1] == f["dense"][1]
        if not all(lv_same) or not cs:
            a, b = f["latency"][0][-1], f["dense"][0][-1]
            diff = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
            print("method", method, name, "levels equal", lv_same, "counters equal", cs, "differing pixels last level", int(diff.sum()), {k: (f["latency"][1][k], f["dense"][1][k]) for k in f["latency"][1] if f["latency"][1][k] != f["dense"][1][k]})
        else:
            print("method", method, name, "ok")
