"""Soak: an 8-partition ctx on one GPU (issue threads, packed gather, bhray_rebalance every 20 frames, bhray_set_partition to arbitrary bounds every 700) renders a pitching,
orbiting camera for N frames; every 250th frame is compared byte for byte with the undivided frame; device memory in use is sampled (the buffers only ever grow to the
frame's size: it must level off)."""
import os, sys, time, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bhusie_amd as B
from tests import common as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tex = T.textures(small=False)
cfg = B.ladder_for_frame((1280, 720), 3, 4)
def uniforms(i):
    a = 0.35 * math.sin(i * 0.004); b = i * 0.002
    cam = B.Camera(position=(19.0 * math.sin(b), 0.0, -19.0 * math.cos(b)), forward=(-math.sin(b) * math.cos(a), math.sin(a), math.cos(b) * math.cos(a)))
    return T.uniforms(integration_method=1 if (i // 3000) % 2 == 0 else 0, time=i / 60.0, camera=cam, model_count=1 if "mesh" in sys.argv[2:] else 0)
def mesh_pos(i):
    return (-10.0 + 6.0 * math.sin(i * 0.01), 2.0 * math.sin(i * 0.003), 30.0), 0 if (i // 400) % 5 == 4 else 1
one = B.RayPass(cfg, device=0); one.set_textures(*tex)
rp = B.RayPass(cfg, devices=[0] * 8, frames_in_flight=22, frames_per_batch=5, speculative_levels=2)
rp.set_textures(*tex)
rng = np.random.default_rng(7)
MESH = "mesh" in sys.argv[2:]
if MESH:            # a mesh that moves every frame (bhray_set_model_transform travels with the frame through the issue threads)
    import tempfile
    from bhusie_amd import assets
    f = tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False); f.write(assets.icosphere_mesh_obj(5, radius=8.0, bump=0.15, seed=3)); f.close()
    model = B.load_model(f.name); os.unlink(f.name)
    one.upload_model(model); rp.upload_model(model)
if len(sys.argv) > 2 and sys.argv[2] == "worst-first":          # every partition holds the whole frame once: after that nothing may grow any more
    rp.set_uniforms(*uniforms(0))
    for p in range(8):
        rp.set_partition([0] * (p + 1) + [720] * (8 - p)); rp.render()
    rp.sync()
mem, checked, applied, t0 = [], 0, 0, time.perf_counter()
for i in range(N):
    if i and i % 20 == 0:
        applied += rp.rebalance()["applied"]
    if i and i % 700 == 0:
        cuts = np.sort(rng.integers(0, 721, size=7)); rp.set_partition([0] + [int(v) for v in cuts] + [720])
    u = uniforms(i)
    if MESH:
        rp.set_model_transform(*mesh_pos(i))
    rp.set_uniforms(*u); rp.render()
    if i % 250 == 0:
        got = rp.read_hdr()
        if MESH:
            one.set_model_transform(*mesh_pos(i))
        one.set_uniforms(*u); one.render(); want = one.read_hdr()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), i
        checked += 1
        free, total = torch.cuda.mem_get_info(0)
        mem.append((total - free) >> 20)
rp.sync()
dt = time.perf_counter() - t0
print(json.dumps({"frames": N, "seconds": round(dt, 1), "frames_compared": checked, "rebalances_applied": applied, "device_MiB_in_use_first_last_max": [mem[0], mem[-1], max(mem)],
                  "device_MiB_samples": mem[:: max(1, len(mem) // 16)]}))
rp.close(); one.close()
