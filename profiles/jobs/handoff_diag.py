import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
import numpy as np
import bhusie_amd as B
from bhusie_amd import assets
tex = (assets.temp_lut(256), assets.reference_disk_texture(1000), assets.sky_texture(4096, 2048, seed=2))
cam, bh = B.Camera(), B.BlackHole()
det = B.RayDetails(integration_method=1, step_size=0.15, max_iterations=2000, angle_division_threshold=0.02, time=0.0)
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
def run(fif, nring, frames=60, do_read=True):
    rp = B.RayPass(cfg, device=0, frames_in_flight=fif, speculative_levels=2)
    rp.set_textures(*tex); rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
    ring = [B.PinnedFrame(1080, 1920) for _ in range(nring)]
    tick = [None] * nring
    for _ in range(fif + 4):
        rp.render()
    rp.sync()
    t_render = t_read = t_wait = 0.0
    t0 = time.perf_counter()
    for i in range(frames):
        k = i % nring
        a = time.perf_counter()
        if tick[k] is not None:
            rp.wait_read(tick[k])
        b = time.perf_counter()
        rp.render()
        c = time.perf_counter()
        if do_read:
            tick[k] = rp.read_hdr_async(ring[k])
        d = time.perf_counter()
        t_wait += b - a; t_render += c - b; t_read += d - c
    for t in tick:
        if t is not None:
            rp.wait_read(t)
    rp.sync()
    dt = time.perf_counter() - t0
    for b_ in ring:
        b_.free()
    rp.close()
    print(json.dumps({"frames_in_flight": fif, "pinned_ring": nring, "read": do_read, "ms_per_frame": round(dt / frames * 1e3, 4),
                      "host_ms_per_frame": {"wait_for_buffer": round(t_wait / frames * 1e3, 4), "bhray_render": round(t_render / frames * 1e3, 4), "bhray_read_hdr_async": round(t_read / frames * 1e3, 4)}}), flush=True)
run(22, 24, do_read=False)
run(22, 24)
run(22, 48)
run(8, 10)
run(4, 6)
run(2, 3)
