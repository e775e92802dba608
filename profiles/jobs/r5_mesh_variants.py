"""One library (BHRAY_LIB), the 1080p RK frame, 400-frame blocks: mesh where the bench has it / mesh culled at the root for every ray / mesh invisible (no-mesh kernel)."""
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
out = []
for label, pos, vis in (("mesh", (-10.0, 0.0, 30.0), 1), ("culled", (0.0, 0.0, -5000.0), 1), ("nomesh", (-10.0, 0.0, 30.0), 0)):
    rp = B.RayPass(cfg, device=0, frames_in_flight=22, speculative_levels=2)
    rp.set_textures(*T.textures(small=False)); rp.upload_model(model); rp.set_model_transform(pos, vis); rp.set_uniforms(*T.uniforms(integration_method=1, model_count=1))
    for _ in range(44): rp.render()
    rp.sync(); best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(400): rp.render()
        rp.sync(); best = min(best, (time.perf_counter() - t0) / 400 * 1e3)
    out.append("%s %.4f" % (label, best))
    rp.close()
print(os.path.basename(os.environ.get("BHRAY_LIB", "default")), " ".join(out), flush=True)
