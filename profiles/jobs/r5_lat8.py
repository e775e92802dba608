"""One frame at a time on a rank of an 8-way partition (frames_in_flight = 1, one frame per batch)."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
import bench
for integ in ("rk", "euler"):
    a = argparse.Namespace(workload="disk", integrator=integ, max_iterations=2000, bvh="reference")
    tex, cam, bh, det, model = bench.build_scene(a)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    for rank in (0, 3):
        rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=2, timing=True, row_rank=rank, row_world=8, slab_row0=[0, 407, 462, 497, 525, 559, 600, 658, 1080])
        rp.set_textures(*tex); rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
        for _ in range(4): rp.render(); rp.sync()
        ts = []
        for _ in range(16):
            t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        tm = rp.timing(); n = max(1, tm.frames)
        print("%-5s rank %d of 8  wall %.3f ms  levels %s" % (integ, rank, sorted(ts)[len(ts) // 2], ["%.3f" % (tm.level_trace_ms[i] / n) for i in range(4)]), flush=True)
        rp.close()
