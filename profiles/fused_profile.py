#!/usr/bin/env python3
"""Where does a fused-ladder frame's time go?  Runs one 1080p frame at a time on the timing-only build (-DBHRAY_EXP_PROFILE) of the same
sources and reads the per-wave phase clocks (bhray_debug_read_profile).  GPU box, from the repo root:
    make -C bhusie_amd/csrc OUT=../../profiles/variants/libbhray_prof.so OBJDIR=_obj_prof EXTRA=-DBHRAY_EXP_PROFILE
    BHRAY_LIB=profiles/variants/libbhray_prof.so python profiles/fused_profile.py [speculative_levels] > gpurun_out/fused_profile.json"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bhusie_amd as B  # noqa: E402
from bhusie_amd import assets  # noqa: E402

spec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
L = C.CDLL(B.LIB_PATH)
L.bhray_debug_read_profile.argtypes = [C.c_void_p, C.c_size_t]
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
rp = B.RayPass(cfg, frames_in_flight=1, fused=True, speculative_levels=spec)
rp.set_textures(*tex); rp.set_uniforms(*u)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); rp.render(); rp.sync(); ts.append(time.perf_counter() - t0)
buf = np.zeros(8192 * 16, np.int64)
L.bhray_debug_read_profile(buf.ctypes.data, buf.size)
d = buf.reshape(8192, 16).astype(float)
TICK_NS = 1.0 / 2.4            # shader clock
tr = d[d[:, 0] > 0]
cl = d[d[:, 13] > 0]
def ms(x): return round(float(x) * TICK_NS * 1e-6, 4)
out = {"frame_ms_host": round(sorted(ts[2:])[2] * 1e3, 4), "speculative_levels": spec, "waves": int(len(tr)), "classifier_waves": int(len(cl)),
       "tracer_mean_ms": {"total": ms(tr[:, 0].mean()), "refill_and_slot_polls": ms(tr[:, 3].mean()), "disk_shading": ms(tr[:, 4].mean()), "flat_phase": ms(tr[:, 5].mean()),
                          "epilogue_and_tile_bookkeeping": ms(tr[:, 6].mean()), "step_batches": ms(tr[:, 7].mean()), "idle_no_rays": ms(tr[:, 12].mean())},
       "tracer_rounds_mean": float(tr[:, 1].mean()), "tracer_steps_mean": float(tr[:, 2].mean()), "tracer_steps_max": float(tr[:, 2].max()),
       "ticks_per_step_in_step_batches": float((tr[:, 7].sum() / max(1.0, tr[:, 2].sum()))),
       "classifier_mean_ms": {"total": ms(cl[:, 13].mean()), "waiting_for_items": ms(cl[:, 14].mean()), "processing": ms((cl[:, 13] - cl[:, 14]).mean())},
       "classifier_items_mean": float(cl[:, 15].mean()), "classifier_us_per_item": round(float(((cl[:, 13] - cl[:, 14]).sum() / max(1.0, cl[:, 15].sum())) * TICK_NS * 1e-3), 3)}
rp.close()
print(json.dumps(out))
