import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhusie_amd as B
from tests import common as T
rng = np.random.default_rng(1234)
tex = T.textures()
bad = 0
for trial in range(40):
    method = int(rng.integers(0, 2))
    pos = rng.normal(size=3) * np.array([12.0, 4.0, 12.0]); pos[2] -= 18.0
    fwd = -pos / np.linalg.norm(pos) + rng.normal(size=3) * 0.15; fwd /= np.linalg.norm(fwd)
    cam = B.Camera(position=tuple(float(v) for v in pos), forward=tuple(float(v) for v in fwd), fov=float(rng.uniform(0.6, 1.6)))
    bh = B.BlackHole(position=tuple(float(v) for v in rng.normal(size=3) * 2.0), relativity_sphere_radius=float(rng.uniform(8.0, 30.0)),
                     accretion_disk_inner=float(rng.uniform(1.5, 4.0)), accretion_disk_outer=float(rng.uniform(5.0, 12.0)))
    u = T.uniforms(integration_method=method, camera=cam, black_hole=bh, step_size=float(rng.uniform(0.05, 0.3)), max_iterations=int(rng.integers(50, 2500)),
                   angle_division_threshold=float(rng.uniform(0.005, 0.08)), time=float(rng.uniform(0, 5)))
    cfg = B.ladder_for_frame((int(rng.integers(60, 420)), int(rng.integers(40, 260))), 3, int(rng.integers(1, 5)))
    frames = {}
    for name, env, kw in (("latency", "0", dict(frames_in_flight=1)), ("dense", "1", dict(frames_in_flight=1)), ("S2 4 slots", None, dict(frames_in_flight=4, speculative_levels=2 if len(cfg.sizes()) >= 3 else 0)),
                          ("temporal", None, dict(frames_in_flight=1, temporal=True) if len(cfg.sizes()) <= 4 else dict(frames_in_flight=1))):
        if env is None: os.environ.pop("BHRAY_TRACE_DENSE", None)
        else: os.environ["BHRAY_TRACE_DENSE"] = env
        rp = B.RayPass(cfg, counters=True, **kw)
        rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
        if name == "temporal": rp.render(); rp.render()
        frames[name] = rp.read_hdr(); rp.close()
    ref = frames["latency"]
    for name, f in frames.items():
        fa, ra = f.copy(), ref.copy(); nn = np.isnan(fa) & np.isnan(ra); fa[nn] = 0; ra[nn] = 0
        same = np.array_equal(fa.view(np.uint32), ra.view(np.uint32))
        if not same:
            bad += 1; print("MISMATCH trial", trial, name, cfg.sizes(), method)
print("trials 40, mismatching frames", bad)
