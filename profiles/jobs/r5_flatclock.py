"""Counting build with BHRAY_EXPERIMENT_FLAT_CLOCK=1, one frame at a time: per ladder level (no speculation), the longest time one wave spent in flat phases
and the number of flat phases, for the mesh where the bench has it and with every ray culled at the root."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
from bhusie_amd import assets
from bhusie_amd.layouts import BhrayCounters
import ctypes as C
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
obj = assets.icosphere_mesh_obj(7, radius=8.0, bump=0.15, seed=3)
f = tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False); f.write(obj); f.close()
model = B.load_model(f.name); os.unlink(f.name)
for label, pos in (("mesh", (-10.0, 0.0, 30.0)), ("culled", (0.0, 0.0, -5000.0))):
    rc = B.RayPass(cfg, device=0, frames_in_flight=1, speculative_levels=0, counters=True)
    rc.set_textures(*T.textures(small=False)); rc.upload_model(model); rc.set_model_transform(pos, 1); rc.set_uniforms(*T.uniforms(integration_method=1, model_count=1)); rc.render()
    for lv in range(4):
        c = BhrayCounters()
        B.check(rc._L.bhray_get_level_counters(rc._h, lv, C.byref(c)), rc._h, rc._L)
        print(label, "level", lv, "traced", c.traced, "flat iterations", c.flat_iters, "wave-steps", c.wave_steps, " flat phases (all waves)", c.rays_adopted, " longest time of ONE wave in flat phases %.3f ms" % (c.max_ray_iterations / 100e3), flush=True)
    rc.close()
