import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
import bench
for integ in ("rk", "euler"):
    a = argparse.Namespace(workload="mesh", integrator=integ, max_iterations=2000, bvh="reference")
    tex, cam, bh, det, model = bench.build_scene(a)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    rc = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=0, counters=True)
    rc.set_textures(*tex); rc.upload_model(model); rc.set_uniforms(cam.uniform(), bh.uniform(), det.uniform()); rc.render()
    c = rc.counters(); sc = rc.scheduling_counters()
    print(integ, "longest traversal (loop iterations):", sc["max_ray_iterations"], " flat iterations", c["flat_iters"], "node pairs", c["node_pairs"], "triangles", c["triangles"], flush=True)

    rc.close()
