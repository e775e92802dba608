"""One 1080p frame at a time: what do the texture fetches of the disk shading cost a launch of nearly empty waves?  The same frame with the disk texture / the temperature LUT
switched off in the uniforms (another picture: the point is the time of the levels)."""
import os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bhusie_amd as B
import bench
for integ in ("rk", "euler"):
    a = argparse.Namespace(workload="disk", integrator=integ, max_iterations=2000, bvh="reference")
    tex, cam, bh, det, model = bench.build_scene(a)
    for name, kw in (("as the bench", {}), ("no disk texture", dict(show_disk_texture=0)), ("no red shift", dict(show_red_shift=0)), ("neither", dict(show_disk_texture=0, show_red_shift=0)),
                     ("no disk at all (inner = outer)", dict(accretion_disk_inner=9.99, accretion_disk_outer=10.0))):
        bh2 = B.BlackHole(**kw)
        cfg = B.ladder_for_frame((1920, 1080), 3, 4)
        rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=2, timing=True)
        rp.set_textures(*tex); rp.set_uniforms(cam.uniform(), bh2.uniform(), det.uniform())
        for _ in range(4): rp.render(); rp.sync()
        ts = []
        for _ in range(16):
            t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        tm = rp.timing(); n = max(1, tm.frames)
        print(f"{integ:6s} {name:32s} wall {sorted(ts)[8]:.3f} ms  trace per level {[round(tm.level_trace_ms[i] / n, 3) for i in range(4)]}")
        rp.close()
