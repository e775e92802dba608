import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhusie_amd as B
from tests import common as T
from oracle import oracle as O
rng = np.random.default_rng(1234)
tex = T.textures()
for trial in range(2):
    method = int(rng.integers(0, 2))
    pos = rng.normal(size=3) * np.array([12.0, 4.0, 12.0]); pos[2] -= 18.0
    fwd = -pos / np.linalg.norm(pos) + rng.normal(size=3) * 0.15; fwd /= np.linalg.norm(fwd)
    cam = B.Camera(position=tuple(float(v) for v in pos), forward=tuple(float(v) for v in fwd), fov=float(rng.uniform(0.6, 1.6)))
    bh = B.BlackHole(position=tuple(float(v) for v in rng.normal(size=3) * 2.0), relativity_sphere_radius=float(rng.uniform(8.0, 30.0)),
                     accretion_disk_inner=float(rng.uniform(1.5, 4.0)), accretion_disk_outer=float(rng.uniform(5.0, 12.0)))
    u = T.uniforms(integration_method=method, camera=cam, black_hole=bh, step_size=float(rng.uniform(0.05, 0.3)), max_iterations=int(rng.integers(50, 2500)),
                   angle_division_threshold=float(rng.uniform(0.005, 0.08)), time=float(rng.uniform(0, 5)))
    cfg = B.ladder_for_frame((int(rng.integers(60, 420)), int(rng.integers(40, 260))), 3, int(rng.integers(1, 5)))
    print("trial", trial, "method", method, cfg.sizes(), "cam", pos, "R", bh.relativity_sphere_radius if hasattr(bh, "relativity_sphere_radius") else "?")
    res = {}
    for name, env in (("latency a", "0"), ("dense", "1"), ("latency b", "0")):
        os.environ["BHRAY_TRACE_DENSE"] = env
        rp = B.RayPass(cfg, counters=True, frames_in_flight=1)
        rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
        res[name] = ([rp.read_level(l) for l in range(len(cfg.sizes()))], rp.counters()); rp.close()
    want = O.render_ladder(T.oracle_scene(*u, tex), cfg.sizes())
    for name, (lv, c) in res.items():
        for l in range(len(lv)):
            d = (lv[l].view(np.uint32) != res["dense"][0][l].view(np.uint32)).any(axis=-1)
            w = want[l]; ok = ~np.isnan(lv[l]).any(axis=-1)
            cls = (lv[l][..., 3][ok] != w[..., 3][ok]).sum()
            with np.errstate(invalid="ignore"):
                err = np.abs(lv[l][ok] - w[ok]) / np.maximum(np.abs(w[ok]), 1e-3)
            print("   ", name, "level", l, "pixels differing from dense:", int(d.sum()), "| vs oracle: class mismatches", int(cls), "max rel err %.3g" % float(np.nanmax(err)))
        if name == "latency a":
            l = len(lv) - 1
            d = (lv[l].view(np.uint32) != res["dense"][0][l].view(np.uint32)).any(axis=-1) if "dense" in res else None
        print("   ", name, "counters", c)
    l = len(cfg.sizes()) - 1
    a, b, w = res["latency a"][0][l], res["dense"][0][l], want[l]
    d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
    ys, xs = np.nonzero(d)
    for y, x in list(zip(ys, xs))[:6]:
        print("    pixel", (x, y), "latency", a[y, x], a[y, x].view(np.uint32), "dense", b[y, x], b[y, x].view(np.uint32), "oracle", w[y, x], w[y, x].view(np.uint32))
