"""The quad march (bhray_quad.inc) against the scalar thin shares: one frame at a time, wall time per frame and trace time per level, for
BHRAY_QUAD = 0 (off) / 1..4 (waves per SIMD it may use); every frame compared byte for byte with the BHRAY_QUAD=0 frame.
BHRAY_QUAD is read at bhray_create, so one process measures all settings on one box.  usage: r6_quad.py [rounds]"""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bhusie_amd as B
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cases = [("disk", "rk", (1920, 1080), 2), ("disk", "euler", (1920, 1080), 2), ("disk", "rk", (1920, 1080), 3), ("disk", "rk", (1920, 1080), 0), ("disk", "rk", (3840, 2160), 2)]
ref = {}
for rnd in range(rounds):
    for wl, integ, size, spec in cases:
        a = argparse.Namespace(workload=wl, integrator=integ, max_iterations=2000, bvh="reference")
        tex, cam, bh, det, model = bench.build_scene(a)
        cfg = B.ladder_for_frame(size, 3, 4)
        for q in (0, 1, 2, 3, 4):
            os.environ["BHRAY_QUAD"] = str(q)
            rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=spec, timing=True)
            rp.set_textures(*tex)
            rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
            for _ in range(4): rp.render(); rp.sync()
            ts = []
            for _ in range(16):
                t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append((time.perf_counter() - t0) * 1e3)
            tm = rp.timing(); n = max(1, tm.frames)
            fr = rp.read_hdr()
            key = (wl, integ, size, spec)
            if q == 0 and key not in ref: ref[key] = fr
            same = np.array_equal(fr.view(np.uint32), ref[key].view(np.uint32))
            print("%-5s %-5s %dx%d S=%d BHRAY_QUAD=%d  wall %.3f ms  levels %s  %s" % (wl, integ, size[0], size[1], spec, q, sorted(ts)[len(ts) // 2],
                  ["%.3f" % (tm.level_trace_ms[i] / n) for i in range(4)], "identical" if same else "DIFFERENT FRAME (%d pixels)" % int((fr.view(np.uint32) != ref[key].view(np.uint32)).any(axis=-1).sum())), flush=True)
            rp.close()
