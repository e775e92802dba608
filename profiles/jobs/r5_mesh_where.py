"""Where the mesh frame's extra time goes: the same scene with the mesh (a) where the bench has it, (b) far away behind the camera - every ray fails the
root-box cull, the mesh VARIANT of the kernel still runs (5 waves per SIMD, flat phases batched, the traversal compiled in) - and (c) invisible (the no-mesh kernel)."""
import os, sys, time, tempfile
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
from bhusie_amd import assets
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
obj = assets.icosphere_mesh_obj(7, radius=8.0, bump=0.15, seed=3)
f = tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False); f.write(obj); f.close()
model = B.load_model(f.name); os.unlink(f.name)
for label, pos, vis in (("mesh at (-10,0,30)", (-10.0, 0.0, 30.0), 1), ("mesh at (0,0,-5000): every ray culled at the root", (0.0, 0.0, -5000.0), 1), ("mesh invisible: the no-mesh kernel", (-10.0, 0.0, 30.0), 0)):
    for counters in (False, True):
        rp = B.RayPass(cfg, device=0, frames_in_flight=22 if not counters else 1, speculative_levels=2, counters=counters)
        rp.set_textures(*T.textures(small=False)); rp.upload_model(model); rp.set_model_transform(pos, vis); rp.set_uniforms(*T.uniforms(integration_method=1, model_count=1))
        if counters:
            rp.render(); c = rp.counters(); s = rp.scheduling_counters()
            print("   counters:", {k: c[k] for k in ("traced", "steps", "flat_iters", "node_pairs", "triangles")}, s, flush=True)
        else:
            for _ in range(44): rp.render()
            rp.sync(); best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(400): rp.render()
                rp.sync(); best = min(best, (time.perf_counter() - t0) / 400 * 1e3)
            print("%-60s %.4f ms per frame (400-frame blocks)" % (label, best), flush=True)
        rp.close()
