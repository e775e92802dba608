import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
for name, kw in (("temporal", dict(temporal=True)), ("S2", dict(speculative_levels=2)), ("S3", dict(speculative_levels=3))):
  for dense in ("0", "1"):
    os.environ["BHRAY_TRACE_DENSE"] = dense
    rp = B.RayPass(cfg, frames_in_flight=1, timing=True, **kw)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    for _ in range(3): rp.render()
    rp.sync(); rp.timing()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
    tm = rp.timing()
    f = tm.frames
    print(name, "dense", dense, "host ms %.3f" % (sorted(ts)[5] * 1e3), "event total %.3f trace %.3f classify %.3f" % (tm.total_ms / f, tm.trace_ms / f, tm.classify_ms / f),
          "level classify", [round(tm.level_classify_ms[i] / f, 3) for i in range(4)], "level trace", [round(tm.level_trace_ms[i] / f, 3) for i in range(4)])
    rp.close()
