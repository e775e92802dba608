"""ONE ctx of N partitions on device 0, warm, then 3 blocks of 20 frames with a pause between them (so that a trace shows them apart)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B
from tests import common as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fpb = int(sys.argv[2]) if len(sys.argv) > 2 else 5
fif = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
bounds8 = [0, 383, 435, 487, 542, 599, 657, 713, 1080]
bounds = {8: bounds8, 4: [0, 435, 542, 657, 1080], 2: [0, 542, 1080]}[n]
rp = B.RayPass(cfg, devices=[0] * n, slab_row0=bounds, frames_in_flight=fif, frames_per_batch=fpb, speculative_levels=2)
rp.set_textures(*T.textures(small=False)); rp.set_uniforms(*T.uniforms(integration_method=1))
for _ in range(fpb * fif * 2): rp.render()
rp.sync()
for rep in range(4):
    time.sleep(0.05)
    t0 = time.perf_counter()
    for _ in range(20): rp.render()
    t1 = time.perf_counter(); rp.sync(); t2 = time.perf_counter()
    print("N %d fpb %d fif %d: block %.3f ms (issue %.3f)" % (n, fpb, fif, (t2 - t0) * 1e3, (t1 - t0) * 1e3), flush=True)
rp.close()
