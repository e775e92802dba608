import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
L = C.CDLL(B.LIB_PATH)
L.bhray_debug_read_profile.argtypes = [C.c_void_p, C.c_size_t]
for base, levels in (((73, 41), 1),):
    cfg = B.ladder_from_base(base, 3, levels)
    u = (B.Camera().uniform(), B.BlackHole().uniform(), B.RayDetails(integration_method=1).uniform())
    rp = B.RayPass(cfg, frames_in_flight=1, timing=True)
    rp.set_textures(*tex); rp.set_uniforms(*u)
    for _ in range(3): rp.render()
    rp.sync(); rp.timing()
    rp.render(); rp.sync()
    tm = rp.timing()
    buf = np.zeros(8192 * 16, np.int64)
    L.bhray_debug_read_profile(buf.ctypes.data, buf.size)
    d = buf.reshape(8192, 16)
    act = d[d[:, 2] > 0]
    print("trace ms", tm.trace_ms / tm.frames, "waves with iterations", len(act))
    order = np.argsort(-act[:, 0])[:8]
    for r in act[order]:
        print("total %d rounds %d iters %d | refill %d (%d) shade %d (%d) flat %d (%d) finish %d (%d) steps %d -> %.0f per iteration" % (r[0], r[1], r[2], r[3], r[8], r[4], r[9], r[5], r[10], r[6], r[11], r[7], r[7] / max(1, r[2])))
    print("mean per-iteration step cost over waves: %.0f ticks; mean total %.0f; max total %d" % ((act[:, 7] / act[:, 2]).mean(), act[:, 0].mean(), act[:, 0].max()))
    if os.environ.get("FINE"):
        it = act[:, 2].astype(float)
        print("FINE per iteration (ticks, incl. ~1 clock read each): loop/between %.0f  integrator %.0f  dist+culls %.0f  branch/tail %.0f" % ((act[:, 8] / it).mean(), (act[:, 9] / it).mean(), (act[:, 10] / it).mean(), (act[:, 11] / it).mean()))
    print("sum of phase means: refill %.0f shade %.0f flat %.0f finish %.0f steps %.0f" % tuple(act[:, 3 + k].mean() for k in range(5)))
