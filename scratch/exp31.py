import os, sys, time, math, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhusie_amd as B
from bhusie_amd import assets, _lib
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
bh, det = B.BlackHole(), B.RayDetails(integration_method=1)
def path(n, da, dy):
    out = []
    for i in range(n):
        a = da * i
        pos = (19.0 * math.sin(a), dy * i, -19.0 * math.cos(a))
        nn = math.sqrt(sum(v * v for v in pos))
        out.append((B.Camera(position=pos, forward=tuple(-v / nn for v in pos)).uniform(), bh.uniform(), det.uniform()))
    return out
L = C.CDLL(B.LIB_PATH)
L.bhray_debug_read_queue.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
os.environ["BHRAY_TEMPORAL_MARGIN"] = "0.5"; os.environ["BHRAY_TEMPORAL_RADIUS"] = "4"
P = path(8, 0.002, 0.03)
rp = B.RayPass(cfg, frames_in_flight=1, temporal=True)
rp.set_textures(*tex)
for i, u in enumerate(P):
    rp.set_uniforms(*u); rp.render(); rp.sync()
    if i >= 6:
        for l in (2, 3):
            buf = np.zeros(4096, np.uint32); n = C.c_uint32(0)
            rc = L.bhray_debug_read_queue(rp._h, l, buf.ctypes.data, 4096, C.byref(n))
            e = buf[:min(n.value, 4096)]
            xs, ys = e & 0x7fff, (e >> 15) & 0x7fff
            lev = rp.read_level(l)
            print("frame", i, "level", l, "misses", n.value, "x range", xs.min(), xs.max(), "y range", ys.min(), ys.max())
            print("   alpha of the missed pixels:", np.unique(lev[ys, xs, 3], return_counts=True))
            pts = sorted(zip(ys.tolist(), xs.tolist()))
            print("   first 40:", pts[:40])
