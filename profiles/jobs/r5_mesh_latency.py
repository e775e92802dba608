"""One frame at a time, mesh variant: where the 2.6 ms are (against 1.2 ms without a mesh).  Per ladder level, the trace launch's duration with
the mesh where the bench has it, with every ray culled at the root (the mesh variant still runs), and with the mesh invisible (no-mesh kernel)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
from bhusie_amd import assets
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
obj = assets.icosphere_mesh_obj(7, radius=8.0, bump=0.15, seed=3)
f = tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False); f.write(obj); f.close()
model = B.load_model(f.name); os.unlink(f.name)
for label, pos, vis in (("mesh", (-10.0, 0.0, 30.0), 1), ("culled", (0.0, 0.0, -5000.0), 1), ("nomesh", (-10.0, 0.0, 30.0), 0)):
    for spec in (2, 0):
        rp = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=spec, timing=True)
        rp.set_textures(*T.textures(small=False)); rp.upload_model(model); rp.set_model_transform(pos, vis); rp.set_uniforms(*T.uniforms(integration_method=1, model_count=1))
        for _ in range(4): rp.render(); rp.sync()
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        tm = rp.timing()
        n = max(1, tm.frames)
        print("%-7s spec %d dense_env %s  wall %.3f ms  trace/frame %.3f  levels %s" % (label, spec, os.environ.get("BHRAY_TRACE_DENSE", "-"), sorted(ts)[len(ts) // 2], tm.trace_ms / n,
              ["%.3f" % (tm.level_trace_ms[i] / n) for i in range(4)]), flush=True)
        rp.close()
    rc = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=0, counters=True)
    rc.set_textures(*T.textures(small=False)); rc.upload_model(model); rc.set_model_transform(pos, vis); rc.set_uniforms(*T.uniforms(integration_method=1, model_count=1))
    rc.render()
    for lv in range(4):
        c = rc.level_counters(lv)
        print("   level", lv, {k: c[k] for k in c if c[k]}, flush=True)
    rc.close()
