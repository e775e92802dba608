"""One frame at a time (frames_in_flight = 1, speculative_levels = 2): wall time per frame and the trace launches per level, for the bench scenes."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
import bench
for wl, integ, size in (("disk", "rk", (1920, 1080)), ("disk", "euler", (1920, 1080)), ("mesh", "rk", (1920, 1080)), ("mesh", "euler", (1920, 1080)), ("disk", "rk", (3840, 2160))):
    a = argparse.Namespace(workload=wl, integrator=integ, max_iterations=2000, bvh="reference")
    tex, cam, bh, det, model = bench.build_scene(a)
    cfg = B.ladder_for_frame(size, 3, 4)
    rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=2, timing=True)
    rp.set_textures(*tex)
    if model is not None: rp.upload_model(model)
    rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
    for _ in range(4): rp.render(); rp.sync()
    ts = []
    for _ in range(16):
        t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    tm = rp.timing(); n = max(1, tm.frames)
    print("%-5s %-5s %dx%d  wall %.3f ms  levels %s" % (wl, integ, size[0], size[1], sorted(ts)[len(ts) // 2], ["%.3f" % (tm.level_trace_ms[i] / n) for i in range(4)]), flush=True)
    rp.close()
