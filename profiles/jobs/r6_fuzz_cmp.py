"""The fuzz scenes of tests/test_gpu_edge_cases.py::test_fuzz_uniform_space rendered by several builds (BHRAY_LIB), every frame compared byte for byte with the first build's.
usage: r6_fuzz_cmp.py libA.so libB.so ..."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
import bhusie_amd as B
from tests import common as T
tex = T.textures()
rng = np.random.default_rng(20260928)
cfg = B.ladder_from_base((20, 12), 3, 2)
out = []
for k in range(40):
    pos = rng.normal(size=3) * np.array([6.0, 4.0, 6.0]) + np.array([0.0, 0.0, -16.0])
    fwd = -pos + rng.normal(size=3) * 4.0
    fwd = fwd / np.linalg.norm(fwd)
    cam = B.Camera(position=tuple(pos), forward=tuple(fwd), fov=float(rng.uniform(0.3, 2.2)))
    inner = float(rng.uniform(1.2, 4.0))
    bh = B.BlackHole(accretion_disk_rotation=tuple(rng.uniform(-1.5, 1.5, size=3)), accretion_disk_inner=inner,
                     accretion_disk_outer=inner + float(rng.uniform(1.0, 12.0)), rotation_speed=float(rng.uniform(0, 10)),
                     relativity_sphere_radius=float(rng.uniform(8.0, 40.0)), show_disk_texture=int(rng.integers(0, 2)),
                     show_red_shift=int(rng.integers(0, 2)), feather_amount=float(rng.uniform(0.05, 1.0)))
    method = int(rng.integers(0, 2))
    u = T.uniforms(camera=cam, black_hole=bh, integration_method=method, step_size=float(rng.uniform(0.05, 0.6)),
                   max_iterations=int(rng.integers(50, 900)), angle_division_threshold=float(rng.uniform(0.0, 0.2)),
                   time=float(rng.uniform(0, 100)))
    rp = B.RayPass(cfg, device=0, speculative_levels=0)
    rp.set_textures(*tex); rp.set_uniforms(*u); rp.render()
    f = rp.read_hdr()
    np.save(sys.argv[1] + "_%%d.npy" %% k, f)
    rp.close()
''' % ROOT
libs = sys.argv[1:]
os.makedirs("gpurun_out/fuzz", exist_ok=True)
import numpy as np
for i, l in enumerate(libs):
    p = subprocess.run([sys.executable, "-c", CHILD, "gpurun_out/fuzz/f%d" % i], capture_output=True, text=True, env=dict(os.environ, BHRAY_LIB=os.path.abspath(l)), timeout=900)
    if p.returncode: print(l, "FAILED", p.stderr[-2000:])
for k in range(40):
    a = np.load("gpurun_out/fuzz/f0_%d.npy" % k)
    for i in range(1, len(libs)):
        b = np.load("gpurun_out/fuzz/f%d_%d.npy" % (i, k))
        same = a.view(np.uint32) == b.view(np.uint32)
        if not same.all():
            bad = np.argwhere(~same.all(axis=-1))
            print("case", k, os.path.basename(libs[i]), "differs in", len(bad), "pixels; first:", bad[:4].tolist())
            for (y, x) in bad[:4]:
                print("   ", a[y, x].tolist(), "vs", b[y, x].tolist())
print("compared", len(libs), "builds over 40 scenes")
for f in os.listdir("gpurun_out/fuzz"): os.remove(os.path.join("gpurun_out/fuzz", f))
