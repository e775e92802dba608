#!/bin/bash
# where do the idle lanes of the step loop come from?  instrumented build (profiles/variants/libbhray_idle.so: counter 11 = lane-steps a lane
# sat EMPTY while its wave's queue was not known to be used up, i.e. lost to the refill threshold); a frame rendered amid 22 in flight
export BHRAY_LIB=$PWD/profiles/variants/libbhray_idle.so
python - <<'P'
import bhusie_amd as B, ctypes as C
from bhusie_amd import layouts
from tests import common as T
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
for fif, fpb in ((1, 1), (22, 1), (22, 2), (8, 8)):
    rp = B.RayPass(cfg, device=0, counters=True, speculative_levels=2, frames_per_batch=fpb, frames_in_flight=fif)
    rp.set_textures(*T.textures(small=False)); rp.set_uniforms(*T.uniforms(integration_method=1, max_iterations=2000))
    for i in range(3 * fif * fpb): rp.render()
    rp.sync()
    c = layouts.BhrayCounters(); B.lib().bhray_get_counters(rp._h, C.byref(c))
    tot = 64.0 * c.wave_steps
    print(f"in flight {fif} batch {fpb}: steps {c.steps} wave_steps {c.wave_steps} marching {c.steps / tot:.4f} empty-refillable {c.rays_adopted / tot:.4f} other idle {(tot - c.steps - c.rays_adopted) / tot:.4f}")
    for l in range(4):
        B.lib().bhray_get_level_counters(rp._h, l, C.byref(c)); t = 64.0 * c.wave_steps
        if t: print(f"     level {l}: steps {c.steps} marching {c.steps / t:.4f} empty-refillable {c.rays_adopted / t:.4f} other {(t - c.steps - c.rays_adopted) / t:.4f}")
    rp.close()
P
