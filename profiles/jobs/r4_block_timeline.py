"""One 20-frame block of rank 3 of an 8-way partition (stripes of 27: no calibration needed) with per-frame time, ladder or temporal:
run under rocprofv3 --kernel-trace; the host sleeps 20 ms before the block so that it stands alone in the trace."""
import sys, time
import bhusie_amd as B
from tests import common as T
mode = sys.argv[1]
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
kw = dict(temporal=True) if mode == "temporal" else dict(speculative_levels=2)
import os
world = int(os.environ.get("TL_WORLD", "8")); fpb = int(os.environ.get("TL_FPB", "10"))
part = dict(row_rank=3, row_world=world) if world > 1 else {}
rp = B.RayPass(cfg, device=0, frames_in_flight=22, frames_per_batch=fpb, **part, **kw)
rp.set_textures(*T.textures(small=False))
us = [T.uniforms(integration_method=1, max_iterations=2000, time=i / 60.0) for i in range(300)]
i = 0
for _ in range(240):
    rp.set_uniforms(*us[i]); rp.render(); i += 1
rp.sync(); time.sleep(0.02)
t0 = time.perf_counter()
for _ in range(20):
    rp.set_uniforms(*us[i]); rp.render(); i += 1
rp.sync()
print(mode, "block ms", (time.perf_counter() - t0) * 1e3)
time.sleep(0.02)
rp.close()
