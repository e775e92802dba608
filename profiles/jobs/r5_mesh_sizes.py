"""Lone launches, mesh variant: is the traversal's time memory latency or instruction issue?  The same sphere at three tessellations (20 480 / 81 920 / 327 680
triangles: 3 / 13 / 52 MB of nodes + leaf records - the first fits one XCD's L2), the extra time of a frame against the same frame with every ray culled at the root,
and (BHRAY_LIB = the LONGEST_TRAVERSAL build) the longest traversal in loop iterations."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
from bhusie_amd import assets
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
longest = "longest" in os.environ.get("BHRAY_LIB", "")
for sub in (5, 6, 7):
    obj = assets.icosphere_mesh_obj(sub, radius=8.0, bump=0.15, seed=3)
    f = tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False); f.write(obj); f.close()
    model = B.load_model(f.name); os.unlink(f.name)
    if longest:
        rc = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=0, counters=True)
        rc.set_textures(*T.textures(small=False)); rc.upload_model(model); rc.set_model_transform((-10.0, 0.0, 30.0), 1); rc.set_uniforms(*T.uniforms(integration_method=1, model_count=1)); rc.render()
        c = rc.counters(); sc = rc.scheduling_counters()
        print("subdivision", sub, "longest traversal", sc["max_ray_iterations"], "node pairs", c["node_pairs"], "triangles", c["triangles"], "flat iterations", c["flat_iters"], flush=True)
        rc.close(); continue
    res = {}
    for label, pos in (("mesh", (-10.0, 0.0, 30.0)), ("culled", (0.0, 0.0, -5000.0))):
        rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=2, timing=True)
        rp.set_textures(*T.textures(small=False)); rp.upload_model(model); rp.set_model_transform(pos, 1); rp.set_uniforms(*T.uniforms(integration_method=1, model_count=1))
        for _ in range(4): rp.render(); rp.sync()
        ts = []
        for _ in range(16):
            t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        tm = rp.timing(); n = max(1, tm.frames)
        res[label] = (sorted(ts)[len(ts) // 2], [tm.level_trace_ms[i] / n for i in range(4)])
        rp.close()
    print("subdivision", sub, "triangles", 20 * 4 ** sub, " wall mesh %.3f culled %.3f ms;  extra per launch %s" % (res["mesh"][0], res["culled"][0],
          ["%.3f" % (a - b) for a, b in zip(res["mesh"][1], res["culled"][1])]), flush=True)
